import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_SHARED = {}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tsdr():
    from tempest_loader import load_package
    return load_package()


@pytest.fixture(scope="session")
def ctx(tsdr):
    """A HIP context.  Fails loudly (never skips) when the library or the device is missing:
    GPU tests must not pass on a fallback."""
    c = tsdr.Context(0)
    _SHARED["ctx"] = c
    return c


@pytest.fixture(autouse=True)
def _fresh_adaptive_route():
    """The session's shared context carries the sync guard's adaptive-route history (which whole buffers run exactly follows
    from the flagged-frame counts of the CALLS BEFORE -- by design a function of the sequence of buffers): a test that compares
    two calls bit for bit must not inherit another test's history.  Restart the window before every test."""
    c = _SHARED.get("ctx")
    if c is not None and getattr(c, "h", None):
        c.set_option("sync_guard_ppb", 20000)
        c.set_option("sync_guard_auto", 1)
    yield


@pytest.fixture(scope="session")
def synth(tsdr):
    import importlib
    return importlib.import_module("tempestsdr_jl_amd.synth")
