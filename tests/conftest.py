import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


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
    return tsdr.Context(0)


@pytest.fixture(scope="session")
def synth(tsdr):
    import importlib
    return importlib.import_module("tempestsdr_jl_amd.synth")
