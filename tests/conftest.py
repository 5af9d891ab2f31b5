import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "single-algebra_amd", "python"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture
def debug_switches(monkeypatch):
    """Route tests.  The SAPCA_* experiment switches exist only in the -DSAPCA_DEBUG_SWITCHES build of the library
    (single-algebra_amd/csrc/switches.h: the release build reads none of them, so a stray variable in a caller's process
    cannot change a route).  For the duration of the test every call of the Python layer goes to that build
    (lib/libsapca_dbg.so: same sources, same kernels), beside the release library the other tests use."""
    from sapca import _lib as L
    L.load()
    monkeypatch.setattr(L, "_lib", L.load_debug())
    yield
