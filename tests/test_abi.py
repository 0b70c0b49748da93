"""The C-ABI library loads without a GPU, exports every symbol include/mvicp.h declares, and refuses to
compute without a device (no CPU fallback anywhere in the product path)."""
import ctypes as C
import os
import re

import pytest

import mvicp
from mvicp import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mvicp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mvicp_[a-z0-9_]+)\s*\(", txt)) - {"mvicp_eval_fn"})


def test_exports_every_declared_symbol(engine_lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(engine_lib, s), f"{s} declared in include/mvicp.h but not exported"
    assert sorted(L.SYMBOLS) == syms, "python binding list out of sync with the header"


def test_version_and_error_string(engine_lib):
    assert b"gfx950" in engine_lib.mvicp_version()
    assert isinstance(engine_lib.mvicp_last_error(), bytes)


def test_no_gpu_means_loud_failure(engine_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    st = engine_lib.mvicp_create(0, C.byref(h))
    assert st == -2 and not h.value
    assert b"no CPU fallback" in engine_lib.mvicp_last_error()
    with pytest.raises(mvicp.MvicpError):
        mvicp.Engine(0)


def test_edge_owner_partition_is_contiguous_and_balanced():
    ns = [200000] * 62
    for world in (1, 2, 4, 8):
        own = L.edge_owner(ns, world)
        assert own[0] == 0 and own[-1] == world - 1
        assert all(b - a in (0, 1) for a, b in zip(own[:-1], own[1:]))
        counts = [int((own == r).sum()) for r in range(world)]
        assert max(counts) - min(counts) <= 1, counts
    own = L.edge_owner([10, 1000, 10, 10], 2)
    assert list(own) == [0, 0, 1, 1] or list(own) == [0, 1, 1, 1]


def test_header_is_plain_c(tmp_path):
    """include/mvicp.h is the drop-in boundary: it must compile as C (C99, pedantic) — no C++-isms, no torch / HIP types in any signature —
    and mvicp_corr must be the reference's 16-byte Correspondance (include/frame.h:18-22)."""
    import subprocess
    src = tmp_path / "abi_c.c"
    src.write_text('#include "mvicp.h"\n#include <stddef.h>\n'
                   'typedef char corr_is_16_bytes[(sizeof(mvicp_corr) == 16 && offsetof(mvicp_corr, second) == 4 && offsetof(mvicp_corr, dist) == 8) ? 1 : -1];\n'
                   'int use(mvicp_ctx* c) { const mvicp_corr* t; const long long* o; return mvicp_map_correspondences(c, &t, &o); }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "abi_c.o")])
