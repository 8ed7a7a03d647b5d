#!/usr/bin/env python3
"""Files -> pixels, end to end, for a batch of baseline 1080p 4:2:0 JPEG files held in HOST memory (not the bench.py
contract, whose inputs are resident in HBM: this is the PCIe- and feeder-inclusive rate DESIGN.md section 6 quotes).
  A. host feeder on T threads (gamut_hip_jpeg_decode_coeffs_batch) -> H2D of the coefficients -> k_jpeg_h2v2
  B. compressed bytes H2D -> k_jpeg_entropy (one lane per image / restart interval) -> k_jpeg_h2v2
Usage: python tools/e2e_bench.py [--batch 256] [--distinct 32] [--restart-rows 0] [--threads 0]"""
import argparse
import ctypes as C
import io
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
from gamut_amd import _capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--restart-rows", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--progressive", action="store_true", help="progressive (SOF2) files: libjpeg's default scan script")
    ap.add_argument("--pinned", action="store_true", help="the files live in page-locked host memory: path C uploads the scans from where they are")
    ap.add_argument("--paths", default="abc", help="which of the three paths to time (a: host feeder, b: device entropy decode, c: files -> pixels)")
    a = ap.parse_args()
    L = _capi.lib()
    _capi.check(L.gamut_hip_init(0))
    w, h, B = 1920, 1080, a.batch
    files = []
    for i in range(a.distinct):
        bio = io.BytesIO()
        kw = dict(quality=90, subsampling=2)
        if a.restart_rows:
            kw["restart_marker_rows"] = a.restart_rows
        if a.progressive:
            kw["progressive"] = True
        Image.fromarray(gen.synth_rgb(w, h, 100 + i)).save(bio, "JPEG", **kw)
        files.append(np.frombuffer(bio.getvalue(), np.uint8))
    if a.pinned:
        total = sum(((f.size + 63) & ~63) for f in files)
        arena = L.gamut_hip_host_malloc_pinned(total)
        assert arena
        whole = np.ctypeslib.as_array(C.cast(arena, C.POINTER(C.c_uint8)), (total,))
        at = 0
        for j, f in enumerate(files):
            whole[at:at + f.size] = f
            files[j] = whole[at:at + f.size]
            at += (f.size + 63) & ~63
    bufs = [files[i % a.distinct] for i in range(B)]
    ptrs = (C.c_void_p * B)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * B)(*[b.size for b in bufs])
    nblk = 120 * 68 * 6
    dev = torch.device("cuda", 0)
    dco = torch.empty((B, nblk * 64), dtype=torch.int16, device=dev)
    dzz = torch.empty((B, nblk), dtype=torch.uint8, device=dev)
    out = torch.empty((B, h, w * 4), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    co_off = (np.arange(B, dtype=np.int64) * nblk * 64); zz_off = (np.arange(B, dtype=np.int64) * nblk)
    pinned = torch.empty((B, nblk * 64), dtype=torch.int16).pin_memory()

    def reconstruct():
        _capi.check(L.gamut_hip_jpeg_reconstruct_batch_device(dco.data_ptr(), nblk * 64, dzz.data_ptr(), nblk, out.data_ptr(), w * 4,
                                                               h * w * 4, w, h, 4, 4, B, stream))

    def path_a():
        frames = (_capi.JpegFrame * B)()
        _capi.check(L.gamut_hip_jpeg_decode_coeffs_batch(ptrs, lens, B, frames, None, a.threads))
        t1 = time.perf_counter()
        for i in range(B):
            C.memmove(pinned[i].data_ptr(), frames[i].coeffs, nblk * 128)
            L.gamut_hip_jpeg_frame_free(C.byref(frames[i]))
        dco.copy_(pinned, non_blocking=True)
        dzz.fill_(64)                                   # dense path (max_zag only matters for the Col!1 corner case)
        reconstruct()
        torch.cuda.synchronize()
        return t1

    def path_b():
        info = (_capi.JpegFrame * B)()
        _capi.check(L.gamut_hip_jpeg_entropy_decode_device(ptrs, lens, B, co_off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                            zz_off.ctypes.data_as(C.POINTER(C.c_int64)), dco.data_ptr(), dzz.data_ptr(), None,
                                                            info, None, stream))
        t1 = time.perf_counter()
        reconstruct()
        torch.cuda.synchronize()
        return t1

    out_off = (np.arange(B, dtype=np.int64) * h * w * 4)

    def path_c():
        info = (_capi.JpegFrame * B)()
        _capi.check(L.gamut_hip_jpeg_decode_batch_device(ptrs, lens, B, 4, out_off.ctypes.data_as(C.POINTER(C.c_int64)), out.data_ptr(), info, None, None, stream))
        return time.perf_counter()

    mb = sum(b.size for b in bufs) / 1e6
    print(f"batch {B} x {w}x{h} {'progressive' if a.progressive else 'baseline'} 4:2:0, {mb / B * 1e3:.0f} kB/file, restart rows {a.restart_rows}, host threads {a.threads or os.cpu_count()}, files in {'PAGE-LOCKED' if a.pinned else 'pageable'} host memory")
    ref = None
    for name, fn in (("A host feeder + coefficient upload", path_a), ("B device entropy decode", path_b), ("C files -> pixels in one call", path_c)):
        if name[0].lower() not in a.paths:
            continue
        best, best_first = 1e9, 0
        for _ in range(a.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t1 = fn()
            dt = time.perf_counter() - t0
            if dt < best:
                best, best_first = dt, t1 - t0
        px = out[B - 1].clone()
        if ref is None:
            ref = px
        same = bool(torch.equal(px, ref))
        print(f"  {name:36s} {B * w * h / best / 1e6:9.1f} Mpx/s   ({best * 1e3:7.1f} ms, of which entropy stage {best_first * 1e3:7.1f} ms)   pixels equal: {same}")


if __name__ == "__main__":
    main()
