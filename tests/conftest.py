import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package

    return load_package()


@pytest.fixture(scope="session")
def wl(pkg):
    import importlib

    return importlib.import_module("c2d_amd.workloads")


@pytest.fixture(scope="session")
def eng(pkg):
    """One Engine (c2d_ctx) on device 0 for the whole GPU session.  No fallback:
    if the HIP library or the device is missing this raises."""
    e = pkg.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu

    cpu.lib()
    return cpu
