#!/usr/bin/env python3
"""PNG files in host memory -> pixels in HBM through gamut_hip_png_decode_batch_device (chunk walk + inflate on host
threads, the rest on the GPU).  Usage: python tools/e2e_png_bench.py [--batch 64] [--width 3840 --height 2160] [--threads 0]"""
import argparse
import ctypes as C
import io
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
from gamut_amd import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=4)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--content", choices=["noisy", "smooth"], default="noisy", help="smooth: the same synthetic image behind a Gaussian blur (3.6 MB per 4K file instead of 12.3)")
    a = ap.parse_args()
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    w, h, B = a.width, a.height, a.batch
    files = []
    for i in range(a.distinct):
        img = Image.fromarray(gen.synth_rgb(w, h, 200 + i))
        if a.content == "smooth":
            from PIL import ImageFilter
            img = img.filter(ImageFilter.GaussianBlur(2))
        bio = io.BytesIO(); img.save(bio, "PNG", compress_level=6)
        files.append(np.frombuffer(bio.getvalue(), np.uint8))
    bufs = [files[i % a.distinct] for i in range(B)]
    ptrs = (C.c_void_p * B)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * B)(*[b.size for b in bufs])
    offs = (np.arange(B, dtype=np.int64) * w * h * 4)
    dout = L.gamut_hip_device_malloc(B * w * h * 4)
    info = (_capi.PngInfo * B)()
    print(f"batch {B} x {w}x{h} RGB8 PNG ({a.content}) -> rgba8, {sum(b.size for b in bufs) / B / 1e6:.1f} MB/file, {os.cpu_count()} host cores")
    for mode, threads in (("host", 1), ("host", 16), ("host", 0), ("device", 0)):
        os.environ["GAMUT_HIP_PNG_INFLATE"] = mode
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            _capi.check(L.gamut_hip_png_decode_batch_device(ptrs, lens, B, 4, 8, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, info, None, threads, None))
            best = min(best, time.perf_counter() - t0)
        print(f"  inflate on the {mode:6s}, host threads {str(threads) if threads else 'auto':>4s}: {B * w * h / best / 1e6:9.1f} Mpx/s  ({best * 1e3:8.1f} ms)")
    L.gamut_hip_device_free(dout)


if __name__ == "__main__":
    main()
