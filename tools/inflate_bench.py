#!/usr/bin/env python3
"""Inflate kernel alone: IDAT streams of N PNG files resident in HBM -> inflated bytes in HBM (gamut_hip_inflate_batch_device),
wall time of launch + sync, against zlib on one host thread.  With a -DINFLATE_PROFILE=1 build (tools/variant.sh
inflate:prof:-DINFLATE_PROFILE=1, GAMUT_HIP_LIB=gamut_amd/lib/var/libgamut_hip_prof.so) the cycles per phase (thread 0's clock, summed over the
streams; h: = header steps, t: = table steps) are printed too.
Usage: python tools/inflate_bench.py [--streams 256] [--width 3840 --height 2160]"""
import argparse
import ctypes as C
import io
import os
import sys
import time
import zlib

import numpy as np
from PIL import Image, ImageFilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
from gamut_amd import _capi  # noqa: E402

PHASES = ["header", "tables", "window", "sweep0", "sweeps", "scan", "emit", "resolve", "flush", "fill", "h:fields", "h:code", "h:walks", "h:emit", "t:ranks", "t:starts", "t:sort", "t:lit", "t:dist", "t:long", "f:setup", "f:init", "s:detect", "r:expand", "r:init", "r:rounds", "s:turn1", "#blocks", "#rounds", "#turns", "#doublings", "#jobs"]
NPH = 27


def idat(png):
    out, p = [], 8
    while p < len(png):
        n = int.from_bytes(png[p:p + 4], "big"); typ = png[p + 4:p + 8]
        if typ == b"IDAT":
            out.append(png[p + 8:p + 8 + n])
        p += 12 + n
    return b"".join(out)[2:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    prof = getattr(C.CDLL(_capi.LIB_PATH), "gamut_hip_inflate_profile", None)
    w, h, N = a.width, a.height, a.streams
    kinds = {}
    rgb = gen.synth_rgb(w, h, 200)
    for name, img, lvl in (("noisy RGB, level 6", Image.fromarray(rgb), 6),
                           ("smooth RGB (blurred), level 6", Image.fromarray(rgb).filter(ImageFilter.GaussianBlur(2)), 6),
                           ("smooth RGB, level 9", Image.fromarray(rgb).filter(ImageFilter.GaussianBlur(2)), 9),
                           ("flat graphics (16 colours), level 6", Image.fromarray((rgb // 64 * 64).astype(np.uint8)).filter(ImageFilter.ModeFilter(9)), 6)):
        bio = io.BytesIO(); img.save(bio, "PNG", compress_level=lvl)
        kinds[name] = idat(bio.getvalue())
    for name, s in kinds.items():
        t0 = time.perf_counter(); raw = zlib.decompressobj(-15).decompress(s); t_host = time.perf_counter() - t0
        comp = np.frombuffer(s, np.uint8)
        stride = (comp.size + 255) & ~255
        dcomp = L.gamut_hip_device_malloc(stride * 1 + 64)                 # every stream reads the same compressed bytes
        _capi.check(L.gamut_hip_memcpy_h2d(dcomp, comp.ctypes.data, comp.size, None))
        cap = len(raw)
        dout = L.gamut_hip_device_malloc((cap + 256) * N)
        dlen = L.gamut_hip_device_malloc(8 * N); dst = dlen + 4 * N
        descs = (_capi.InflateDesc * N)()
        for i in range(N):
            descs[i].src = dcomp; descs[i].dst = dout + i * (cap + 256); descs[i].src_len = comp.size; descs[i].dst_cap = cap
        _capi.check(L.gamut_hip_stream_synchronize(None))
        best = 1e9
        times = []
        for rep in range(a.reps):
            if prof:
                buf = (C.c_ulonglong * len(PHASES))(); prof(buf, 1)
            poison = np.full(2 * N, 0xFFFFFFFF, np.uint32)                 # every repetition must write its own lengths and verdicts
            _capi.check(L.gamut_hip_memcpy_h2d(dlen, poison.ctypes.data, 8 * N, None))
            _capi.check(L.gamut_hip_stream_synchronize(None))
            t0 = time.perf_counter()
            _capi.check(L.gamut_hip_inflate_batch_device(descs, N, dlen, dst, None))
            _capi.check(L.gamut_hip_stream_synchronize(None))
            times.append(time.perf_counter() - t0)
            chk = np.zeros(2 * N, np.uint32)
            _capi.check(L.gamut_hip_memcpy_d2h(chk.ctypes.data, dlen, 8 * N, None))
            _capi.check(L.gamut_hip_stream_synchronize(None))
            if not ((chk[N:] == 0).all() and (chk[:N] == cap).all()):
                print(f"  repetition {rep}: {int((chk[:N] != cap).sum())} streams without their length, {int((chk[N:] != 0).sum())} without a verdict of 0 (first words {chk[:2]}, {chk[N:N + 2]}) in {times[-1] * 1e3:.2f} ms")
        best = min(times)
        print("  repetitions: " + ", ".join(f"{x * 1e3:.1f} ms" for x in times[:8]))
        st = np.zeros(2 * N, np.uint32)
        _capi.check(L.gamut_hip_memcpy_d2h(st.ctypes.data, dlen, 8 * N, None))
        got = np.empty(cap, np.uint8)
        _capi.check(L.gamut_hip_memcpy_d2h(got.ctypes.data, dout + (N - 1) * (cap + 256), cap, None))
        _capi.check(L.gamut_hip_stream_synchronize(None))
        ok = (st[N:] == 0).all() and (st[:N] == cap).all() and got.tobytes() == raw
        print(f"{name}: {comp.size / 1e6:.2f} MB -> {cap / 1e6:.2f} MB; zlib on one host thread {cap / t_host / 1e6:.0f} MB/s")
        print(f"  {N} streams: {best * 1e3:.1f} ms, {N * cap / best / 1e9:.1f} GB/s inflated ({N * w * h / best / 1e6:.0f} Mpx/s), parity {'ok' if ok else 'FAIL'}")
        if prof:
            buf = (C.c_ulonglong * len(PHASES))(); prof(buf, 0)
            v = list(buf); tot = sum(v[:NPH]) or 1
            print("  cycles: " + ", ".join(f"{PHASES[k]} {100 * v[k] / tot:.1f}%" for k in range(NPH) if v[k]))
            print("  per stream: " + ", ".join(f"{PHASES[k]} {v[k] / N:.0f}" for k in range(NPH, len(PHASES))) + f"; {tot / N / 1e6:.1f} Mcycles")
        for p in (dcomp, dout, dlen):
            L.gamut_hip_device_free(p)


if __name__ == "__main__":
    main()
