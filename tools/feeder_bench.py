#!/usr/bin/env python3
"""Host feeder throughput: gamut_hip_jpeg_decode_coeffs_batch on N threads over copies of one synthetic 1080p JPEG
(baseline and progressive), in Mpixels/s.  Host-only (no GPU needed).  Usage: python tools/feeder_bench.py [threads...]"""
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
    L = _capi.lib()
    w, h, copies = 1920, 1080, 64
    img = Image.fromarray(gen.synth_rgb(w, h, 3))
    threads = [int(a) for a in sys.argv[1:]] or [1, 2, 4, os.cpu_count()]
    for prog in (False, True):
        bio = io.BytesIO(); img.save(bio, "JPEG", quality=90, subsampling=2, progressive=prog)
        buf = np.frombuffer(bio.getvalue(), np.uint8)
        ptrs = (C.c_void_p * copies)(*[buf.ctypes.data] * copies)
        lens = (C.c_size_t * copies)(*[buf.size] * copies)
        for t in threads:
            dt = 1e9
            for _ in range(3):                       # best of 3: the first pass pays the page faults of fresh heap memory
                frames = (_capi.JpegFrame * copies)()
                t0 = time.perf_counter()
                _capi.check(L.gamut_hip_jpeg_decode_coeffs_batch(ptrs, lens, copies, frames, None, t))
                dt = min(dt, time.perf_counter() - t0)
                for i in range(copies):
                    L.gamut_hip_jpeg_frame_free(C.byref(frames[i]))
            print(f"{'progressive' if prog else 'baseline':11s} {buf.size / 1e3:7.1f} kB/file  threads={t:3d}  {copies * w * h / dt / 1e6:8.1f} Mpx/s")


if __name__ == "__main__":
    main()
