import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import gen, oracle_lib as O
from gamut_amd import _capi
from test_png_gpu import gpu_defilter
L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
img_n, out_n, color = 4, 4, 6
rng = np.random.default_rng(7)
x, y, n = 64, 1000, 2
stride = (x * img_n + 1) * y + 5
raws = np.zeros(n * stride, np.uint8); exps = []
for i in range(n):
    f = rng.integers(0, 5, y).astype(np.uint8)
    r = gen.png_forward_filter(rng.integers(0, 256, (y, x * img_n)), img_n, f)
    raws[i * stride:i * stride + r.size] = r
    exps.append(O.png_create_image_raw(r, img_n, out_n, x, y, 8, color).reshape(y, -1))
got = gpu_defilter(L, raws, x, y, img_n, out_n, 8, color, count=n, raw_stride=stride)
for i in range(n):
    g = got[i].reshape(y, -1)
    bad = np.nonzero((g != exps[i]).any(axis=1))[0]
    print("img", i, "bad rows", len(bad), bad[:4], bad[-4:] if len(bad) else "")
