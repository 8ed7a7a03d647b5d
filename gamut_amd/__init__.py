"""gamut_amd -- MI355X (gfx950) batched image-decode / pixel-convert path for Gamut.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of
include/gamut_hip.h, built into `lib/libgamut_hip.so`) and thin Python access
to that ABI for tests and bench.py.  There is no CPU fallback.
"""
from . import _capi  # noqa: F401

__all__ = ["_capi"]
