import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # a fresh checkout has no binaries (they are git-ignored): build the library and the oracle once (make is incremental;
    # hipcc cross-compiles gfx950 without a GPU).  On the GPU box the snapshot already carries them.
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (os.path.exists(os.path.join(root, "gamut_amd", "lib", "libgamut_hip.so")) and os.path.exists(os.path.join(root, "oracle", "liboracle.so"))):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def hip():
    """The C-ABI library, initialised on cuda:0.  GPU tests call through this."""
    from gamut_amd import _capi
    L = _capi.lib()
    _capi.check(L.gamut_hip_init(0))
    return L
