import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gen, oracle_lib as O
from gamut_amd import _capi
from gamut_amd.image import Image
import gamut_amd.image as gi
import test_image_gpu as T
L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
rng = np.random.default_rng(4)
w, h = 21, 13
G = T.G
files = {"issue76": open(os.path.join(G, "ref_images", "issue76.png"), "rb").read(),
         "vst3": open(os.path.join(G, "ref_images", "vst3-compatible.png"), "rb").read(),
         "rgb8": gen.write_png(rng.integers(0, 256, (h, w * 3)), w, h, 2, 8),
         "rgba16": gen.write_png(rng.integers(0, 65536, (h, w * 4)), w, h, 6, 16),
         "pal4": gen.write_png(rng.integers(0, 16, (h, w)), w, h, 3, 4, palette=rng.integers(0, 256, (16, 3)), trns=[0, 128, 255]),
         "la8": gen.write_png(rng.integers(0, 256, (h, w * 2)), w, h, 4, 8, interlace=1)}
for name, data in files.items():
    for flags in T.FLAGSETS + [gi.LOAD_8BIT, gi.LOAD_16BIT]:
        ew, eh, t1, exp = T.expected_load(data, flags, "png")
        for rep in range(2):
            im = Image()
            ok = im.loadFromMemory(data, flags | gi.LAYOUT_TRAILING[3])
            got = im.pixels()
            if got.shape != exp.shape: print(name, hex(flags), "shape", got.shape, exp.shape); continue
            bad = np.argwhere(got != exp)
            if len(bad): print(name, hex(flags), rep, "mismatches", len(bad), bad[:10].tolist(), got[tuple(bad[0])], exp[tuple(bad[0])])
print("done")
