import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mv-lm-icp_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/_build/liborc.so), built on demand.  Test infrastructure only."""
    import orclib
    return orclib.load()


@pytest.fixture(scope="session")
def refnn():
    """The real vendored nanoflann behind oracle/_ref/libref_nanoflann.so, or None if it was never built."""
    import orclib
    return orclib.load_ref()


@pytest.fixture(scope="session")
def engine_lib():
    import mvicp
    return mvicp.load_library()
