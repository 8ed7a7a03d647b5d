#!/usr/bin/env python3
"""Mutation fuzzing of gamut_hip_png_decode_batch_device on the GPU box: batches of damaged PNG files decoded twice -- inflate on the
host threads (zlib) and inflate on the GPU (sliced launches behind the upload) -- must agree file by file: same verdict, same pixels,
same info; and both must come back -- and with the oracle's stbi_load (stbdec.d) on the same bytes: same verdict, same pixels.  Usage: python tools/fuzz_png_gpu.py [batches=20] [seed=1]"""
import ctypes as C
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import oracle_lib as O  # noqa: E402
from gamut_amd import _capi  # noqa: E402


def seeds(rng):
    out = []
    for k in range(24):
        w, h = 40 + 23 * (k % 7), 30 + 17 * (k % 5)
        rgb = gen.synth_rgb(w, h, 500 + k)
        mode = k % 6
        if mode == 0: im = Image.fromarray(rgb)
        elif mode == 1: im = Image.fromarray(np.dstack([rgb, rgb[:, :, 0]]))
        elif mode == 2: im = Image.fromarray(rgb[:, :, 1])
        elif mode == 3: im = Image.fromarray((rgb[:, :, 0].astype(np.uint16) * 257 + rgb[:, :, 2]).astype(np.uint16))
        elif mode == 4: im = Image.fromarray(rgb).quantize(40)
        else: im = Image.fromarray(rgb)
        bio = io.BytesIO(); im.save(bio, "PNG", compress_level=(1, 6, 9)[k % 3])
        out.append(bio.getvalue())
    big = gen.synth_rgb(900, 700, 77)                                            # a stream of several slices' worth would need MBs: one mid-size file
    bio = io.BytesIO(); Image.fromarray(big).save(bio, "PNG", compress_level=6); out.append(bio.getvalue())
    return out


def mutate(rng, f):
    s = bytearray(f)
    kind = int(rng.integers(0, 6))
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))): s[int(rng.integers(8, len(s)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1: s = s[:int(rng.integers(8, len(s)))]
    elif kind == 2:
        i = int(rng.integers(8, len(s))); n = int(rng.integers(1, 40)); s[i:i + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 3:                                                               # damage inside the IDAT data
        i = s.find(b"IDAT")
        if i > 0:
            j = i + 4 + int(rng.integers(0, max(1, len(s) - i - 20))); s[min(j, len(s) - 1)] ^= 1 << int(rng.integers(0, 8))
    elif kind == 4: pass                                                          # intact
    else:
        i = s.find(b"IDAT")
        if i > 0: s[i + 4 + int(rng.integers(0, 3))] ^= 0x21                      # zlib header / first block header
    return bytes(s)


def decode(L, files, mode):
    os.environ["GAMUT_HIP_PNG_INFLATE"] = mode
    n = len(files)
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs])
    room = 900 * 700 * 8 + 64
    offs = (np.arange(n, dtype=np.int64) * room)
    dout = L.gamut_hip_device_malloc(n * room)
    fill = np.full(n * room, 0x3C, np.uint8)
    _capi.check(L.gamut_hip_memcpy_h2d(dout, fill.ctypes.data, fill.size, None))
    info = (_capi.PngInfo * n)(); st = (C.c_int * n)()
    L.gamut_hip_png_decode_batch_device(ptrs, lens, n, 4, 8, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, info, st, 4, None)
    host = np.empty(n * room, np.uint8)
    _capi.check(L.gamut_hip_memcpy_d2h(host.ctypes.data, dout, host.size, None)); _capi.check(L.gamut_hip_stream_synchronize(None))
    L.gamut_hip_device_free(dout)
    return [int(v) for v in st], [(i.width, i.height) if not s else None for i, s in zip(info, st)], host.reshape(n, room)


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    base = seeds(rng)
    bad = ok = rejected = 0
    for b in range(batches):
        files = [mutate(rng, base[int(rng.integers(0, len(base)))]) for _ in range(40)]
        s_h, i_h, p_h = decode(L, files, "host")
        s_d, i_d, p_d = decode(L, files, "device")
        for k in range(len(files)):
            if (s_h[k] == 0) != (s_d[k] == 0) or i_h[k] != i_d[k]:
                bad += 1; print(f"MISMATCH batch {b} file {k}: host status {s_h[k]} {i_h[k]}, device status {s_d[k]} {i_d[k]}")
            elif s_h[k] == 0:
                w, h = i_h[k]
                if not np.array_equal(p_h[k][: w * h * 4], p_d[k][: w * h * 4]):
                    bad += 1; print(f"MISMATCH batch {b} file {k}: pixels differ")
                else:
                    ok += 1
                    r = O.stbi_load(files[k], 4, False)                           # ... and against the oracle's stbi_load (8-bit rgba): verdict and pixels
                    want = None if r is None else np.ascontiguousarray(r[0]).view(np.uint8).reshape(-1)
                    if want is None:
                        bad += 1; print(f"MISMATCH batch {b} file {k}: decoded, the oracle rejects it")
                    elif want.size != w * h * 4 or not np.array_equal(want, p_d[k][: w * h * 4]):
                        bad += 1; print(f"MISMATCH batch {b} file {k}: pixels differ from the oracle's")
                        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True); open(os.path.join(ROOT, "gpurun_out", f"fuzz_png_mismatch_{b}_{k}.png"), "wb").write(files[k])
            else:
                rejected += 1
                if O.stbi_load(files[k], 4, False) is not None:
                    bad += 1; print(f"MISMATCH batch {b} file {k}: rejected (status {s_d[k]}), the oracle decodes it")
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True); open(os.path.join(ROOT, "gpurun_out", f"fuzz_png_mismatch_{b}_{k}.png"), "wb").write(files[k])
    print(f"fuzz_png_gpu: {batches} batches of 40 files: {ok} decoded alike by both inflaters, {rejected} rejected by both, {bad} mismatches")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
