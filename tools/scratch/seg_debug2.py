import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import gen, oracle_lib as O
from gamut_amd import _capi
from test_png_gpu import gpu_defilter
L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
rng = np.random.default_rng(1)
x, y = 64, 1000
px = rng.integers(0, 256, (y, x * 4))
filt = np.full(y, 3, np.uint8); filt[507] = 1
raw = gen.png_forward_filter(px, 4, filt)
exp = O.png_create_image_raw(raw, 4, 4, x, y, 8, 6).reshape(y, -1)
got = gpu_defilter(L, raw, x, y, 4, 4, 8, 6)[0].reshape(y, -1)
print("bad rows", np.count_nonzero((got != exp).any(axis=1)))
