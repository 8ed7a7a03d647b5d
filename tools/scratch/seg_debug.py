import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import gen, oracle_lib as O
from gamut_amd import _capi
from test_png_gpu import gpu_defilter
L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
img_n, out_n, color = 4, 4, 6
rng = np.random.default_rng(40 + img_n)
for (x, y) in [(97, 256), (64, 1000), (301, 2050)]:
    px = rng.integers(0, 256, (y, x * img_n))
    patterns = [np.full(y, 4, np.uint8), np.full(y, 2, np.uint8), rng.integers(0, 5, y).astype(np.uint8), rng.integers(2, 5, y).astype(np.uint8)]
    one = np.full(y, 3, np.uint8); one[y // 2 + 7] = 1; patterns.append(one)
    far = rng.integers(2, 5, y).astype(np.uint8); far[3] = 0; far[y - 2] = 1; far[y // 8 + y // 20] = 1; patterns.append(far)
    for k, filt in enumerate(patterns):
        raw = gen.png_forward_filter(px, img_n, filt)
        exp = O.png_create_image_raw(raw, img_n, out_n, x, y, 8, color).reshape(y, -1)
        for rep in range(3):
            got = gpu_defilter(L, raw, x, y, img_n, out_n, 8, color)[0].reshape(y, -1)
            bad = np.nonzero((got != exp).any(axis=1))[0]
            pois = np.nonzero((got == 0xA5).all(axis=1))[0]
            if len(bad): print(x, y, "pattern", k, "rep", rep, "bad rows", len(bad), bad[:5], bad[-5:], "poison rows", len(pois), "cut rows:", np.nonzero(filt <= 1)[0][:12])
print("done")
