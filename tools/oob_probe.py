#!/usr/bin/env python3
"""Out-of-bounds probe for the device entry points (GPU AddressSanitizer is not available on this pool): every input and
output buffer of a call is placed so that it ENDS exactly at the end of its own hipMalloc allocation (a multiple of 2 MiB), so
a kernel that reads or writes past a buffer runs into unmapped memory and the process aborts with a memory access fault.
`python tools/oob_probe.py control` does such an over-read on purpose (exit code != 0 expected) to show the probe can see it.
Usage: python tools/oob_probe.py [control]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
from gamut_amd import _capi  # noqa: E402

L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
CHUNK = 2 << 20
VERBOSE = bool(os.environ.get('OOB_VERBOSE'))
allocs = []


def at_end(nbytes, align=1):
    """device pointer to nbytes whose last byte is the last byte of a fresh allocation"""
    total = (max(nbytes, 1) + CHUNK - 1) // CHUNK * CHUNK
    p = L.gamut_hip_device_malloc(total)
    assert p
    allocs.append(p)
    q = p + total - nbytes
    assert q % align == 0, (nbytes, align)
    return q


def up_end(arr, align=1):
    arr = np.ascontiguousarray(arr)
    q = at_end(arr.nbytes, align)
    _capi.check(L.gamut_hip_memcpy_h2d(q, arr.ctypes.data, arr.nbytes, None)); _capi.check(L.gamut_hip_stream_synchronize(None))
    return q


def free_all():
    _capi.check(L.gamut_hip_stream_synchronize(None))
    for p in allocs:
        L.gamut_hip_device_free(p)
    allocs.clear()


def png_cases():
    rng = np.random.default_rng(1)
    n = 0
    for img_n, depth, color in [(1, 1, 0), (1, 8, 0), (1, 16, 0), (2, 8, 4), (3, 8, 2), (3, 16, 2), (4, 8, 6), (4, 16, 6)]:
        fb = 1 if depth < 8 else img_n * (2 if depth == 16 else 1)
        for x, y in [(1, 1), (5, 2), (7, 5), (37, 1), (37, 3), (64, 65), (130, 66), (259, 71), (1000, 9), (33, 129)]:
            rows = gen.pack_samples(rng.integers(0, 1 << depth, (y, x * img_n)), depth)
            raw = gen.png_forward_filter(rows, fb, rng.integers(0, 5, y).astype(np.uint8))
            for out_n in ([img_n] + ([img_n + 1] if img_n in (1, 3) else [])):
                nbytes = x * y * out_n * (2 if depth == 16 else 1)
                if VERBOSE: print('png', img_n, depth, color, x, y, out_n, flush=True)
                draw = up_end(raw)
                dout = at_end(nbytes)
                dst = at_end(4, 4)
                _capi.check(L.gamut_hip_png_defilter_batch_device(draw, 0, raw.size, dout, nbytes, x, y, img_n, out_n, depth, color, 1, dst, None))
                free_all(); n += 1
    return n


def jpeg_cases():
    rng = np.random.default_rng(2)
    NB = {0: 1, 1: 3, 2: 4, 3: 4, 4: 6}; MCU = {0: (8, 8), 1: (8, 8), 2: (16, 8), 3: (8, 16), 4: (16, 16)}
    n = 0
    for st in (0, 1, 2, 3, 4):
        for w, h in [(1, 1), (17, 9), (128, 16), (129, 17), (250, 33), (1920, 24)]:
            mw, mh = MCU[st]
            nblk = ((w + mw - 1) // mw) * ((h + mh - 1) // mh) * NB[st]
            co = rng.integers(-200, 200, (nblk, 64)).astype(np.int16)
            for oc in (4, 3, 1):
                if VERBOSE: print('jpeg', st, w, h, oc, flush=True)
                dco = up_end(co, 16)
                dout = at_end(w * h * oc)
                _capi.check(L.gamut_hip_jpeg_reconstruct_batch_device(dco, nblk * 64, None, 0, dout, w * oc, w * h * oc, w, h, st, oc, 1, None))
                free_all(); n += 1
    return n


def convert_cases(extra_rows=0):
    rng = np.random.default_rng(3)
    size = [1, 2, 4, 2, 4, 8, 2, 4, 8, 3, 6, 12, 4, 8, 16, 4, 8, 16]
    n = 0
    for s in range(18):
        for d in range(18):
            for w, h in [(1, 1), (7, 3), (64, 2), (1001, 3)]:
                src = rng.integers(0, 256, w * h * size[s]).astype(np.uint8)
                if size[s] in (4, 8, 12, 16) and s % 3 == 2:                    # f32 types: finite values
                    src = rng.random(w * h * size[s] // 4, dtype=np.float32).view(np.uint8)
                if VERBOSE: print('convert', s, d, w, h, flush=True)
                ds_ = up_end(src)
                dd = at_end(w * h * size[d])
                _capi.check(L.gamut_hip_scanlines_convert_device(s, ds_, w * size[s], 0, d, dd, w * size[d], 0, w, h + extra_rows, 1, None))
                free_all(); n += 1
            if extra_rows:
                return n
    return n


if len(sys.argv) > 1 and sys.argv[1] == "control":
    print("control: converting one row more than the buffers hold -- a memory access fault is the expected outcome", flush=True)
    L.gamut_hip_device_free(L.gamut_hip_device_malloc(64 << 20))
    n = 0
    rng = np.random.default_rng(3)
    w, h = 1 << 18, 8                                                     # 1 MiB rows: the extra rows are far outside the allocation
    src = rng.integers(0, 256, w * h * 4).astype(np.uint8)
    ds_ = up_end(src); dd = at_end(w * h * 4)
    _capi.check(L.gamut_hip_scanlines_convert_device(12, ds_, w * 4, 0, 12 + 3, dd, w * 4, 0, w, h + 64, 1, None))
    _capi.check(L.gamut_hip_stream_synchronize(None))
    print("control: NO fault (the probe is blind on this box)")
    sys.exit(0)

print("png", png_cases(), "cases ok", flush=True)
print("jpeg", jpeg_cases(), "cases ok", flush=True)
print("convert", convert_cases(), "cases ok", flush=True)
print("oob_probe: no access outside any buffer")
