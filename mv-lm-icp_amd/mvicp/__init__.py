"""Python-side harness for libmvicp_hip.so (tests / bench only — the product is the C-ABI library).

`lib`   : ctypes binding of include/mvicp.h (one class, `Engine`, mirroring the call order of the
          reference's main_multiview.cpp loop).
`synth` : the build-owned deterministic synthetic multi-view generator.
`io`    : the reference's on-disk formats (.xyz clouds, 4x4 row-major pose text).
"""
from . import lib, synth, io  # noqa: F401
from .lib import Engine, MvicpError, load_library  # noqa: F401
