#!/usr/bin/env python3
"""BASELINE.json config 5, end to end on one GPU: a mixed batch of 1080p files in HOST memory (image i: JPEG / PNG / QOI by
i % 3) -> rgba8 pixels in HBM, through the three file-level batch entry points.  Per-format wall times (the three calls run
one after the other).  Not the bench.py contract (inputs there are resident in HBM): this is the feeder- and PCIe-inclusive rate.
Usage: python tools/e2e_mixed_bench.py [--batch 96] [--distinct 8]"""
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
from gamut_amd import _capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--pinned", action="store_true", help="the files live in page-locked host memory (gamut_hip_host_malloc_pinned): the JPEG and QOI "
                    "legs then upload them from where they are, without a staging copy")
    a = ap.parse_args()
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    w, h, B, D = 1920, 1080, a.batch, a.distinct
    imgs = [gen.synth_rgb(w, h, 300 + i) for i in range(D)]
    enc = {"jpeg": [], "png": [], "qoi": []}
    for im in imgs:
        b = io.BytesIO(); Image.fromarray(im).save(b, "JPEG", quality=90, subsampling=2); enc["jpeg"].append(np.frombuffer(b.getvalue(), np.uint8))
        b = io.BytesIO(); Image.fromarray(im).save(b, "PNG", compress_level=6); enc["png"].append(np.frombuffer(b.getvalue(), np.uint8))
        enc["qoi"].append(np.frombuffer(synth.qoi_encode(im), np.uint8))
    if a.pinned:                                                # one page-locked arena, the distinct files back to back (64-byte aligned)
        total = sum(((f.size + 63) & ~63) for k in enc for f in enc[k])
        arena = L.gamut_hip_host_malloc_pinned(total)
        assert arena
        whole = np.ctypeslib.as_array(C.cast(arena, C.POINTER(C.c_uint8)), (total,))
        at = 0
        for k in enc:
            for j, f in enumerate(enc[k]):
                whole[at:at + f.size] = f
                enc[k][j] = whole[at:at + f.size]
                at += (f.size + 63) & ~63
    kinds = ["jpeg", "png", "qoi"]
    idx = {k: [i for i in range(B) if kinds[i % 3] == k] for k in kinds}
    out = torch.empty((B, h * w * 4), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    img = h * w * 4

    def arrays(k):
        bufs = [enc[k][(i // 3) % D] for i in idx[k]]
        n = len(bufs)
        return bufs, n, (C.c_void_p * n)(*[b.ctypes.data for b in bufs]), (np.array(idx[k], np.int64) * img)

    nblk = 120 * 68 * 6
    jb, nj, jp, joff = arrays("jpeg"); jl = (C.c_size_t * nj)(*[b.size for b in jb])
    dco = torch.empty((nj, nblk * 64), dtype=torch.int16, device="cuda"); dzz = torch.empty((nj, nblk), dtype=torch.uint8, device="cuda")
    co_off = np.arange(nj, dtype=np.int64) * nblk * 64; zz_off = np.arange(nj, dtype=np.int64) * nblk
    pb, npn, pp, poff = arrays("png"); pl = (C.c_size_t * npn)(*[b.size for b in pb])
    qb, nq, qp, qoff = arrays("qoi"); ql = (C.c_int * nq)(*[b.size for b in qb])
    P64 = C.POINTER(C.c_int64)

    def run_jpeg():
        info = (_capi.JpegFrame * nj)()
        _capi.check(L.gamut_hip_jpeg_entropy_decode_device(jp, jl, nj, co_off.ctypes.data_as(P64), zz_off.ctypes.data_as(P64), dco.data_ptr(), dzz.data_ptr(), None, info, None, stream))
        # the JPEG images of the batch sit at every third slot: image pitch = 3 slots
        _capi.check(L.gamut_hip_jpeg_reconstruct_batch_device(dco.data_ptr(), nblk * 64, dzz.data_ptr(), nblk, out.data_ptr(), w * 4, 3 * img, w, h, 4, 4, nj, stream))
        torch.cuda.synchronize()

    def run_png():
        info = (_capi.PngInfo * npn)()
        _capi.check(L.gamut_hip_png_decode_batch_device(pp, pl, npn, 4, 8, poff.ctypes.data_as(P64), out.data_ptr(), info, None, a.threads, stream))

    def run_qoi():
        descs = (_capi.QoiDesc * nq)()
        _capi.check(L.gamut_hip_qoi_decode_batch_device(qp, ql, nq, 4, qoff.ctypes.data_as(P64), out.data_ptr(), descs, None, stream))

    # the same batch through ONE call: gamut_hip_decode_batch_device sniffs the formats and runs the three pipelines side by side
    allb = [enc[kinds[i % 3]][(i // 3) % D] for i in range(B)]
    ap_ = (C.c_void_p * B)(*[b.ctypes.data for b in allb]); al_ = (C.c_size_t * B)(*[b.size for b in allb])
    aoff = np.arange(B, dtype=np.int64) * img

    def run_all():
        info = (_capi.ImageInfo * B)()
        _capi.check(L.gamut_hip_decode_batch_device(ap_, al_, B, 4, aoff.ctypes.data_as(P64), out.data_ptr(), info, None, stream))

    print(f"mixed batch of {B} x {w}x{h} files ({nj} JPEG {np.mean([b.size for b in jb]) / 1e3:.0f} kB, {npn} PNG {np.mean([b.size for b in pb]) / 1e6:.1f} MB, "
          f"{nq} QOI {np.mean([b.size for b in qb]) / 1e6:.1f} MB) "
          f"-> rgba8 in HBM; {a.threads} host threads (0 = all; PNG batches this large inflate on the GPU); "
          f"files in {'PAGE-LOCKED' if a.pinned else 'pageable'} host memory")
    best = {}
    for rep in range(3):
        for k, fn in (("jpeg", run_jpeg), ("png", run_png), ("qoi", run_qoi)):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            best[k] = min(best.get(k, 1e9), time.perf_counter() - t0)
    sep = out.clone()
    out.zero_()
    t_all = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run_all(); torch.cuda.synchronize()
        t_all = min(t_all, time.perf_counter() - t0)
    same = bool(torch.equal(sep, out))
    del sep
    got = out.cpu().numpy().reshape(B, h, w, 4)
    ok = all(np.array_equal(got[i, :, :, :3], imgs[(i // 3) % D]) for i in idx["png"][:2] + idx["qoi"][:2]) and (got[:, :, :, 3] == 255).all()
    ref = np.array(Image.open(io.BytesIO(enc["jpeg"][0].tobytes())).convert("RGB")).astype(int)
    ok = ok and np.abs(got[0, :, :, :3].astype(int) - ref).max() <= 40            # another decoder (libjpeg-turbo upsamples differently): sanity only
    for k, n in (("jpeg", nj), ("png", npn), ("qoi", nq)):
        print(f"  {k:4s} {n:4d} files  {best[k] * 1e3:9.1f} ms  {n * w * h / best[k] / 1e6:10.1f} Mpx/s")
    tot = sum(best.values())
    print(f"  all  {B:4d} files  {tot * 1e3:9.1f} ms  {B * w * h / tot / 1e6:10.1f} Mpx/s   pixels {'ok' if ok else 'MISMATCH'}   (the three calls one after the other)")
    print(f"  ONE CALL (gamut_hip_decode_batch_device: formats sniffed, pipelines side by side)  {t_all * 1e3:9.1f} ms  {B * w * h / t_all / 1e6:10.1f} Mpx/s   "
          f"pixels {'== the three calls' if same else 'DIFFER from the three calls'}")


if __name__ == "__main__":
    main()
