#!/usr/bin/env python3
"""make_scanline_vectors.py -- golden vectors for the PixelType conversion matrix, produced by EXECUTING the reference's own
scanline functions (tools/d_scanline_exec.py transpiles source/gamut/scanline.d statement by statement; needs
/root/reference, so it runs in the build container only).  Output: tests/golden/scanline_ref.npz

    in_<type>            input pixels of one row (uint8 view), chosen per source type:
                           8-bit:   every value of a channel against every value of alpha (premultiplied types), ramps otherwise
                           16-bit:  a stride through all 65536 codes, 0 / 1 / 65535 / mid codes, random alpha pairs
                           f32:     k/255 and k/65535 grid points, the rounding boundaries (k + 0.5)/M -+ 1 ulp, 0, 1, random [0,1)
    out_<src>_<dst>      what scanlinesConvert gives for in_<src> -> <dst>, all 18 x 18 pairs, through the reference's
                         dispatch (scanlinesInterType, convertToIntermediateScanline, convertFromIntermediate)
    sha_*                sha256 of exhaustive sweeps too large to store: every 16-bit code through u16 -> f32 -> u16 / u8
The npz is data (inputs + expected outputs), not source text.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from d_scanline_exec import Reference  # noqa: E402

TYPES = ["l8", "l16", "lf32", "la8", "la16", "laf32", "lap8", "lap16", "lapf32", "rgb8", "rgb16", "rgbf32",
         "rgba8", "rgba16", "rgbaf32", "rgbap8", "rgbap16", "rgbapf32"]
SIZE = dict(zip(TYPES, [1, 2, 4, 2, 4, 8, 2, 4, 8, 3, 6, 12, 4, 8, 16, 4, 8, 16]))
CH = dict(zip(TYPES, [1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4]))
N = 1024                                   # pixels per stored row


def inputs(rng):
    out = {}
    for t in TYPES:
        ch = CH[t]
        if t.endswith("f32"):
            k8 = rng.integers(0, 256, N * ch); k16 = rng.integers(0, 65536, N * ch)
            v = rng.random(N * ch, dtype=np.float32)
            sel = rng.integers(0, 6, N * ch)
            b8 = ((k8 + 0.5) / 255.0).astype(np.float32); b16 = ((k16 + 0.5) / 65535.0).astype(np.float32)
            v = np.where(sel == 1, (k8 / np.float32(255.0)).astype(np.float32), v)
            v = np.where(sel == 2, (k16 / np.float32(65535.0)).astype(np.float32), v)
            v = np.where(sel == 3, np.nextafter(b8, np.float32(rng.choice([0, 1]))), v)
            v = np.where(sel == 4, np.nextafter(b16, np.float32(rng.choice([0, 1]))), v)
            v = v.astype(np.float32)
            v[:8 * ch] = np.tile(np.array([0, 1, 0.5, 1 / 3, 2 / 3, 0.25, 0.999999, 1e-7], np.float32), ch)
            if "p" in t[1:] and ch in (2, 4):                      # premultiplied source: colour <= alpha mostly, alpha = 0 and tiny alphas present
                a = v.reshape(N, ch)[:, -1].copy(); a[8:40] = 0; a[40:72] = np.float32(1e-3)
                px = v.reshape(N, ch); px[:, -1] = a
                px[72:, :-1] *= a[72:, None]
                v = px.reshape(-1)
            out[t] = v.astype(np.float32).view(np.uint8)
        elif t.endswith("16"):
            v = rng.integers(0, 65536, N * ch).astype(np.uint16)
            v[:16 * ch] = np.repeat(np.array([0, 1, 2, 127, 128, 255, 256, 257, 32767, 32768, 32769, 65279, 65280, 65533, 65534, 65535], np.uint16), ch)
            if "p" in t[1:]:
                px = v.reshape(N, ch); px[16:48, -1] = 0; px[48:80, -1] = 1
                px[80:, :-1] = (px[80:, :-1].astype(np.uint32) * px[80:, -1:].astype(np.uint32) // 65535).astype(np.uint16)
                px[80:400, :-1] = rng.integers(0, 65536, (320, ch - 1))            # also colour > alpha (not a valid premultiplied pixel, but defined)
                v = px.reshape(-1)
            out[t] = v.view(np.uint8)
        else:
            v = rng.integers(0, 256, N * ch).astype(np.uint8)
            v[:256 * ch] = np.repeat(np.arange(256, dtype=np.uint8), ch)             # every code, all channels equal
            if "p" in t[1:]:
                px = v.reshape(N, ch)
                px[256:384, -1] = 0; px[384:512, -1] = 1; px[512:640, -1] = 255
                px[640:, :-1] = (px[640:, :-1].astype(np.uint32) * px[640:, -1:].astype(np.uint32) // 255).astype(np.uint8)
                v = px.reshape(-1)
            out[t] = v
    return out


def build():
    R = Reference()
    rng = np.random.default_rng(20260930)
    arrays = {}
    ins = inputs(rng)
    for t, v in ins.items():
        arrays["in_" + t] = v
    for s in TYPES:
        for d in TYPES:
            if s == d:
                continue
            arrays[f"out_{s}_{d}"] = R.convert_row(s, d, ins[s], N, SIZE)
    # exhaustive sweeps, hashed
    sha = {}
    all16 = np.arange(65536, dtype=np.uint16)
    all8 = np.arange(256, dtype=np.uint8)
    for s, d, v in [("l16", "lf32", all16), ("l16", "l8", all16), ("l8", "l16", all8), ("l8", "lf32", all8), ("l16", "rgba8", all16), ("l8", "rgbaf32", all8)]:
        sha[f"{s}_{d}"] = hashlib.sha256(R.convert_row(s, d, v, v.size, SIZE).tobytes()).hexdigest()
    c, a = np.meshgrid(all8, all8, indexing="ij")
    pairs = np.stack([c, a], axis=-1).reshape(-1).astype(np.uint8)                      # every (colour, alpha) pair
    for s, d in [("lap8", "laf32"), ("lap8", "la8"), ("la8", "lap8"), ("lap8", "rgbaf32"), ("la8", "lap16")]:
        sha[f"pairs_{s}_{d}"] = hashlib.sha256(R.convert_row(s, d, pairs, 65536, SIZE).tobytes()).hexdigest()
    arrays["sha_json"] = np.frombuffer(json.dumps(sha, sort_keys=True).encode(), np.uint8)
    return arrays


if __name__ == "__main__":
    arrays = build()
    path = os.path.join(ROOT, "tests", "golden", "scanline_ref.npz")
    np.savez_compressed(path, **arrays)
    print(path, os.path.getsize(path), "bytes,", len(arrays), "arrays")
