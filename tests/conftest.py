import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_available():
    import ctypes

    from pygradflow_amd import _lib

    cnt = ctypes.c_int(0)
    rc = _lib.load().pgf_device_count(ctypes.byref(cnt))
    return rc == 0 and cnt.value > 0


@pytest.fixture(scope="session")
def pgf():
    import pygradflow_amd as pgf

    return pgf
