#!/usr/bin/env python3
"""Single-image latency of the host drop-in calls (BASELINE.json config 1: one 640x480 baseline JPEG -> rgba8 through
Image.loadFromMemory), next to Pillow (libjpeg-turbo / libpng) decoding the same bytes on one CPU core.  Not a throughput
number: PCIe, launch latency and the host-side entropy decode / inflate dominate."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gamut_amd import image as gi  # noqa: E402
from gamut_amd.image import Image  # noqa: E402


def timeit(fn, reps=30):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e3, min(ts) * 1e3


def main():
    G = os.path.join(ROOT, "tests", "golden")
    cases = [("cfg1 640x480 4:2:0 jpeg -> rgba8", open(os.path.join(G, "jpeg", "cfg1_640x480_420_q90.jpg"), "rb").read(), gi.LOAD_RGB | gi.LOAD_ALPHA | gi.LOAD_8BIT, "jpeg"),
             ("issue65.png -> as stored", open(os.path.join(G, "ref_images", "issue65.png"), "rb").read(), 0, "png")]
    import io
    import gen
    from PIL import Image as PImage
    bio = io.BytesIO(); PImage.fromarray(gen.synth_rgb(1920, 1080, 5)).save(bio, "JPEG", quality=90, subsampling=2)
    cases.append(("1920x1080 4:2:0 jpeg -> rgb8", bio.getvalue(), 0, "jpeg3"))
    bio = io.BytesIO(); PImage.fromarray(gen.synth_rgb(3840, 2160, 6)).save(bio, "PNG", compress_level=1)
    cases.append(("3840x2160 rgb8 png -> rgb8", bio.getvalue(), 0, "png"))
    for name, data, flags, kind in cases:
        im = Image()
        med, best = timeit(lambda: im.loadFromMemory(data, flags))
        dim = Image(device=True)
        dmed, dbest = timeit(lambda: dim.loadFromMemory(data, flags))
        assert im.isValid if hasattr(im, "isValid") else True
        cmed, cbest = timeit(lambda: np.asarray(PImage.open(io.BytesIO(data)).convert("RGBA" if kind == "jpeg" else "RGB")), reps=5)
        print(f"{name:40s} {im.width}x{im.height}: GPU drop-in median {med:7.3f} ms (best {best:7.3f}), pixels left in HBM {dmed:7.3f} ms   Pillow, 1 CPU core: {cmed:7.3f} ms")


def convert_case():
    import ctypes as C
    from gamut_amd import _capi
    L = _capi.lib()
    w = h = 4096
    src = np.random.default_rng(1).integers(0, 256, w * h * 4, dtype=np.uint8)
    dst = np.empty(w * h * 16, np.uint8)
    fn = lambda: _capi.check(L.gamut_hip_scanlines_convert(12, src.ctypes.data, w * 4, 14, dst.ctypes.data, w * 16, w, h))      # rgba8 -> rgbaf32
    med, best = timeit(fn, reps=5)
    t0 = time.perf_counter(); (src[: w * 4 * 256].astype(np.float32) / np.float32(255.0)); c = (time.perf_counter() - t0) * h / 256 * 1e3
    print(f"{'scanlinesConvert rgba8->rgbaf32':40s} {w}x{h}: GPU drop-in median {med:7.3f} ms (best {best:7.3f})   numpy (x / 255.0f), 1 CPU core: {c:7.3f} ms")


if __name__ == "__main__":
    main()
    convert_case()

