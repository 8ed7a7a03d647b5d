import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def hip():
    """The C-ABI library, initialised on cuda:0.  GPU tests call through this."""
    from gamut_amd import _capi
    L = _capi.lib()
    _capi.check(L.gamut_hip_init(0))
    return L
