#!/usr/bin/env python3
"""make_jpeg_vectors.py -- golden vectors of the H2V2 JPEG path, produced IN THE BUILD CONTAINER by the reference-derived
restatement tools/ref_literal_jpeg.py (every arithmetic statement transliterated mechanically from jpegload.d by
tools/make_ref_literal.py) and committed as data: tests/golden/jpeg_h2v2_ref.npz.

    python tools/make_jpeg_vectors.py            (re)write the file
    python tools/make_jpeg_vectors.py --check    exit 1 if the committed file differs from a fresh run

What the GPU box gets is this file, not the tool: `-m gpu` tests compare the HIP kernels -- and liboracle.so -- with these
bytes directly (tests/test_jpeg_gpu.py::test_reference_derived_vectors, tests/test_oracle_on_gpu_box.py).

Contents (all little-endian numpy arrays):
  blocks (K, 64) int16, block_max_zag (K,) uint8      chroma-style inputs: natural / dense 11-bit / wild int16 incl. +-32767
      extremes, coefficients beyond max_zag zeroed (what decode_next_row leaves), every max_zag class of s_max_rc and of the
      row / column tables
  expanded (K, 4, 64) int16                            transform_mcu_expand's four blocks per input (after cast(short), transposed store)
  samples4 (K, 4, 64) uint8                            idct_4x4 of each
  idct (K, 64) uint8                                   idct(block, block_max_zag): DC shortcut, Row!N / Col!N incl. the Col!1 quirk
  frame{i}_coeffs / _max_zag / _w / _h / _rgba         whole 4:2:0 frames (ragged sizes), rgba8 as decompress_jpeg_image_from_stream
      produces it with req_comps = 4 (transform_mcu_expand + expanded_convert per scanline, cropped); frames with max_zag = 64
      everywhere are the dense case (max_zag NULL at the kernel boundary)
rgb8 / l8 are derived by the tests from rgba8 exactly as :3776-3792 does (drop A; (R*19595 + G*38470 + B*7471 + 32768) >> 16)."""
import importlib.util
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "tests", "golden", "jpeg_h2v2_ref.npz")


def literal():
    spec = importlib.util.spec_from_file_location("ref_literal_jpeg", os.path.join(ROOT, "tools", "ref_literal_jpeg.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def blocks_of(rng, n, zag):
    b = np.zeros((n, 64), np.int16)
    k = n // 3
    scale = 1024.0 / (1 + np.arange(64)) ** 1.2
    nat = (rng.standard_normal((k, 64)) * scale).round().clip(-1024, 1023) * (rng.random((k, 64)) < 0.5)
    b[:k][:, zag] = nat.astype(np.int16)
    b[k:2 * k] = rng.integers(-1024, 1024, (k, 64))
    b[2 * k:] = rng.integers(-32768, 32768, (n - 2 * k, 64))
    ext = b[2 * k:2 * k + 8]
    ext[:] = np.where(rng.random(ext.shape) < 0.5, 32767, -32768)
    return b


def build():
    R = literal()
    zag = np.array(R.g_ZAG)
    rng = np.random.default_rng(20260930)
    out = {}
    classes = [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 14, 15, 16, 20, 21, 22, 27, 28, 29, 35, 36, 43, 50, 57, 63, 64]
    bl, mz = [], []
    for c in classes:
        b = blocks_of(rng, 12, zag)
        b[:, zag[c:]] = 0
        bl.append(b); mz += [c] * 12
    blocks = np.concatenate(bl); mz = np.array(mz, np.uint8)
    expanded = np.zeros((len(blocks), 4, 64), np.int16); samples4 = np.zeros((len(blocks), 4, 64), np.uint8); idct = np.zeros((len(blocks), 64), np.uint8)
    for c in classes:
        sel = mz == c
        t, s = R.chroma_expand(blocks[sel], c)
        expanded[sel] = np.moveaxis(t, 0, 1); samples4[sel] = np.moveaxis(s, 0, 1)
        idct[sel] = R.idct(blocks[sel], c)
    out.update(blocks=blocks, block_max_zag=mz, expanded=expanded, samples4=samples4, idct=idct)
    frames = [(48, 32, "zag"), (40, 23, "zag"), (17, 1, "dense"), (1, 33, "zag"), (64, 16, "dense"), (130, 20, "sparse")]
    for i, (w, h, kind) in enumerate(frames):
        mr, mc = (w + 15) // 16, (h + 15) // 16
        co = blocks_of(rng, mr * mc * 6, zag)
        if kind == "dense":
            fz = np.full(mr * mc * 6, 64, np.uint8)
        else:
            fz = rng.choice([1, 2, 3, 6, 10, 11, 21, 36, 64], mr * mc * 6).astype(np.uint8)
            if kind == "sparse":                                 # whole MCU pairs with every Y block <= 10: the kernel's wave-uniform Row!4 / Col!4 passes
                fz = fz.reshape(mc, mr, 6)
                fz[:, : (mr // 2) * 2 - 2, :4] = rng.choice([1, 2, 5, 10], (mc, (mr // 2) * 2 - 2, 4))
                fz = fz.reshape(-1)
            for k in range(len(co)):
                co[k, zag[fz[k]:]] = 0
        rgba = R.decode_h2v2_rgba(co.reshape(mr * mc, 6, 64), fz if kind != "dense" else None, w, h)
        out[f"frame{i}_coeffs"] = co; out[f"frame{i}_max_zag"] = fz; out[f"frame{i}_w"] = np.int32(w); out[f"frame{i}_h"] = np.int32(h)
        out[f"frame{i}_dense"] = np.bool_(kind == "dense"); out[f"frame{i}_rgba"] = rgba
    out["n_frames"] = np.int32(len(frames))
    return out


def main():
    data = build()
    if "--check" in sys.argv:
        cur = np.load(DST)
        bad = [k for k in data if k not in cur.files or not np.array_equal(cur[k], data[k])]
        if bad or set(cur.files) != set(data):
            raise SystemExit(f"{DST}: differs from a fresh run of the reference-derived restatement in {bad or 'its key set'}")
        print("jpeg_h2v2_ref.npz is current")
        return
    bio = io.BytesIO()
    np.savez_compressed(bio, **data)
    open(DST, "wb").write(bio.getvalue())
    print(f"wrote {DST}: {len(bio.getvalue())} bytes, {len(data['blocks'])} blocks, {int(data['n_frames'])} frames")


if __name__ == "__main__":
    main()
