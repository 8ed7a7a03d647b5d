#!/usr/bin/env python3
"""What a caller with FILES sees: batches of files in host memory -> pixels in HBM through the batch entry points (PCIe and the
feeders included -- not bench.py's contract, whose inputs are resident in HBM).  One JSON line per case, for bench.py's `also` array:
  jpeg        1024 x 1080p baseline 4:2:0 files -> rgba8   (gamut_hip_jpeg_decode_batch_device: entropy decode + reconstruction on the GPU)
  jpeg:progressive   the same pictures as progressive (SOF2) files: ten scans each, all of them decoded on the GPU
  png:noisy   256 x 4K RGB8 files of the synthetic image as it is (12 MB of IDAT each)   } gamut_hip_png_decode_batch_device: chunk walk on the
  png:smooth  the same image behind a Gaussian blur (3.6 MB each)                        } host, inflate + de-filter + expansion on the GPU
Parity before timing: JPEG == the oracle's whole-file decoder (oracle/oracle_jpeg.c), every pixel; PNG == Pillow's decode of the same
file, byte for byte (the PNG oracle's inflate is zlib: tests/test_oracle_pinning.py pins it against Pillow on the same files).  Usage: python tools/files_bench.py [jpeg png:noisy png:smooth] [--reps 3]"""
import argparse
import ctypes as C
import io
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image, ImageFilter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
from gamut_amd import _capi  # noqa: E402


def best_of(fn, reps):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def jpeg_case(L, reps, progressive=False):
    w, h, B, distinct = 1920, 1080, 1024, 8
    files = []
    for i in range(distinct):
        bio = io.BytesIO(); Image.fromarray(gen.synth_rgb(w, h, 100 + i)).save(bio, "JPEG", quality=90, subsampling=2, progressive=progressive)
        files.append(np.frombuffer(bio.getvalue(), np.uint8))
    bufs = [files[i % distinct] for i in range(B)]
    ptrs = (C.c_void_p * B)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * B)(*[b.size for b in bufs])
    dev = torch.device("cuda", 0)
    out = torch.empty((B, h, w * 4), dtype=torch.uint8, device=dev)
    off = (np.arange(B, dtype=np.int64) * h * w * 4)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        info = (_capi.JpegFrame * B)()
        _capi.check(L.gamut_hip_jpeg_decode_batch_device(ptrs, lens, B, 4, off.ctypes.data_as(C.POINTER(C.c_int64)), out.data_ptr(), info, None, None, stream))
    run(); torch.cuda.synchronize()
    # parity: the ORACLE's whole-file decoder (oracle/oracle_jpeg.c: decompress_jpeg_image_from_stream restated, entropy decode included)
    # on every distinct file, against the first, a middle and the last image that was decoded from it
    import oracle_lib as O
    ok = True
    for i in range(distinct):
        exp = O.decompress_jpeg(bytes(files[i]), 4)[0].reshape(h, w * 4)
        for j in sorted({i, i + distinct * ((B // distinct) // 2), i + B - distinct}):
            ok = ok and np.array_equal(out[j].cpu().numpy(), exp)
    t = best_of(run, reps)
    kind = "progressive (libjpeg's ten-scan script; every scan decoded on the GPU in one launch)" if progressive else "baseline"
    return {"what": f"files -> pixels: 1024 x 1080p {kind} JPEG 4:2:0 files in host memory -> rgba8 in HBM (gamut_hip_jpeg_decode_batch_device)",
            "value": round(B * w * h / t / 1e6, 1), "unit": "Mpx/s", "ms": round(t * 1e3, 2), "MB_per_file": round(sum(f.size for f in files) / distinct / 1e6, 3),
            "parity": f"ok (== oracle's decompress_jpeg, {3 * distinct} images of the batch, every pixel)" if ok else "FAILED"}


def png_case(L, reps, content):
    w, h, B, distinct = 3840, 2160, 256, 2
    files, refs = [], []
    for i in range(distinct):
        img = Image.fromarray(gen.synth_rgb(w, h, 200 + i))
        if content == "smooth":
            img = img.filter(ImageFilter.GaussianBlur(2))
        bio = io.BytesIO(); img.save(bio, "PNG", compress_level=6)
        files.append(np.frombuffer(bio.getvalue(), np.uint8))
        refs.append(np.asarray(Image.open(io.BytesIO(bio.getvalue())).convert("RGBA")))
    bufs = [files[i % distinct] for i in range(B)]
    ptrs = (C.c_void_p * B)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * B)(*[b.size for b in bufs])
    offs = (np.arange(B, dtype=np.int64) * w * h * 4)
    out = torch.empty((B, h, w, 4), dtype=torch.uint8, device=torch.device("cuda", 0))
    info = (_capi.PngInfo * B)()

    def run():
        _capi.check(L.gamut_hip_png_decode_batch_device(ptrs, lens, B, 4, 8, offs.ctypes.data_as(C.POINTER(C.c_int64)), out.data_ptr(), info, None, 0, None))
    run(); torch.cuda.synchronize()
    ok = all(np.array_equal(out[i].cpu().numpy(), refs[i % distinct]) for i in (0, 1, B - 1))
    t = best_of(run, reps)
    return {"what": f"files -> pixels: 256 x 3840x2160 RGB8 PNG files ({content} content) in host memory -> rgba8 in HBM (gamut_hip_png_decode_batch_device: inflate on the GPU, launched slice by slice behind the upload)",
            "value": round(B * w * h / t / 1e6, 1), "unit": "Mpx/s", "ms": round(t * 1e3, 2), "MB_per_file": round(sum(f.size for f in files) / distinct / 1e6, 2),
            "parity": "ok (== Pillow's decode of the files)" if ok else "FAILED"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=["jpeg", "jpeg:progressive", "png:noisy", "png:smooth"])
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    bad = False
    for c in a.cases:
        r = jpeg_case(L, a.reps) if c == "jpeg" else jpeg_case(L, a.reps, True) if c == "jpeg:progressive" else png_case(L, a.reps, c.split(":")[1])
        r["inputs"] = "files in host memory: PCIe and the feeders are inside the time (bench.py's `value` has its inputs resident in HBM)"
        bad = bad or not r["parity"].startswith("ok")
        print(json.dumps(r), flush=True)
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
