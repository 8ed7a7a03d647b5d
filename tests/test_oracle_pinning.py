"""CPU tests: pin the oracle (the CPU restatement of the reference) against everything independent we have:
the reference's own fixtures and known answers, Pillow-derived golden hashes (tests/golden/golden.json, made by
tools/make_fixtures.py), a float model of the frequency-domain upsample, a numpy binary32 model of the
conversions, and the losslessness of PNG."""
import fixtures
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

import gen
import oracle_lib as O
from oracle_lib import PIXEL_TYPES, PT, PT_CHANNELS, PT_DTYPE, PT_SIZE

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
GOLDEN = json.load(open(os.path.join(G, "golden.json")))
ZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42,
       49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------ PNG
@pytest.mark.parametrize("name", ["issue65.png", "vst3-compatible.png", "issue76.png", "issue92-no-IEND.png", "issue92-truncated-in-CRC.png"])
def test_png_reference_fixtures_match_independent_decoder(name):
    """PNG is lossless: the oracle must reproduce what Pillow decoded (hash committed in golden.json)."""
    data = open(os.path.join(G, "ref_images", name), "rb").read()
    arr, n = O.stbi_load(data, 0, sixteen=(name == "issue76.png"))
    g = GOLDEN["pillow"][name]
    assert list(np.squeeze(arr).shape) == g["shape"]
    assert sha(np.squeeze(arr)) == g["sha"]


def test_png_issue76_known_answer():
    """examples/test-suite/source/main.d:172-190: 2x2 l16 -> 1875, 65535, 0, 2807"""
    arr, n = O.stbi_load(open(os.path.join(G, "ref_images", "issue76.png"), "rb").read(), 0, sixteen=True)
    assert n == 1 and arr.reshape(-1).tolist() == [1875, 65535, 0, 2807]


def test_png_inflate_known_answer():
    """examples/test-suite/source/main.d:51-70: the captured zlib chunk inflates to 594825 + 272 bytes"""
    assert len(zlib.decompress(open(os.path.join(G, "ref_images", "buggy-miniz-chunk.bin"), "rb").read())) == 594825 + 272
    info = O.png_parse(open(os.path.join(G, "ref_images", "vst3-compatible.png"), "rb").read())
    assert info["interlace"] == 1 and (info["width"], info["height"]) == (481, 309)


def test_png_cgbi_frozen():
    """Apple CgBI files: headerless inflate, no BGRA swap / un-premultiply (stbdec.d:1864-1867, 1815); Pillow rejects them."""
    for f in ["issue51cgbi.png", "issue51cgbi2.png"]:
        arr, n = O.stbi_load(open(os.path.join(G, "ref_images", f), "rb").read())
        g = GOLDEN["frozen"][f]
        assert list(arr.shape) == g["shape"] and n == g["comps"] and sha(arr) == g["sha"]


@pytest.mark.parametrize("img_n,depth", [(1, 1), (1, 2), (1, 4), (1, 8), (1, 16), (2, 8), (2, 16), (3, 8), (3, 16), (4, 8), (4, 16)])
def test_png_defilter_roundtrip_all_filters(img_n, depth):
    """forward-filter (independent numpy code) then oracle de-filter gives the samples back, every filter type,
    first-row / first-pixel special cases included (stbdec.d:1381-1388, 1453-1465)."""
    rng = np.random.default_rng(depth + img_n)
    fb = 1 if depth < 8 else img_n * (2 if depth == 16 else 1)
    for (x, y) in [(1, 1), (3, 2), (17, 9), (64, 5)]:
        smp = rng.integers(0, 1 << depth, (y, x * img_n))
        rows = gen.pack_samples(smp, depth)
        for filt in [np.full(y, t, np.uint8) for t in range(5)] + [rng.integers(0, 5, y).astype(np.uint8)]:
            raw = gen.png_forward_filter(rows, fb, filt)
            out = O.png_create_image_raw(raw, img_n, img_n, x, y, depth)
            scale = {1: 255, 2: 85, 4: 17}.get(depth, 1) if img_n == 1 else 1
            if depth == 16:
                assert np.array_equal(out.view(np.uint16), smp.reshape(-1).astype(np.uint16))
            else:
                assert np.array_equal(out, (smp.reshape(-1) * scale).astype(np.uint8))
            if img_n in (1, 3):
                out2 = O.png_create_image_raw(raw, img_n, img_n + 1, x, y, depth)
                w16 = out2.view(np.uint16 if depth == 16 else np.uint8).reshape(y * x, img_n + 1)
                assert np.array_equal(w16[:, :img_n].reshape(-1), (out.view(np.uint16) if depth == 16 else out))
                assert (w16[:, img_n] == (65535 if depth == 16 else 255)).all()
    assert O.png_create_image_raw(np.array([7, 1, 2, 3, 4], np.uint8), 4, 4, 1, 1, 8) is None        # filter > 4: corrupt
    assert O.png_create_image_raw(np.array([0, 1, 2], np.uint8), 4, 4, 1, 1, 8) is None              # not enough pixels


def test_png_generated_files_vs_source_pixels():
    rng = np.random.default_rng(5)
    w, h = 19, 11
    smp = rng.integers(0, 256, (h, w * 4))
    for interlace in (0, 1):
        arr, n = O.stbi_load(gen.write_png(smp, w, h, 6, 8, interlace=interlace), 0)
        assert n == 4 and np.array_equal(arr.reshape(h, w * 4), smp.astype(np.uint8))
    pal = rng.integers(0, 256, (16, 3))
    idx = rng.integers(0, 16, (h, w))
    arr, n = O.stbi_load(gen.write_png(idx, w, h, 3, 4, palette=pal), 0)
    assert n == 3 and np.array_equal(arr, pal[idx].astype(np.uint8))
    g16 = rng.integers(0, 65536, (h, w))
    assert np.array_equal(O.stbi_load(gen.write_png(g16, w, h, 0, 16), 0, sixteen=True)[0][:, :, 0], g16.astype(np.uint16))
    assert np.array_equal(O.stbi_load(gen.write_png(g16, w, h, 0, 16), 0)[0][:, :, 0], (g16 >> 8).astype(np.uint8))      # LOAD_8BIT path: >> 8 (stbdec.d:645)
    g8 = rng.integers(0, 256, (h, w))
    assert np.array_equal(O.stbi_load(gen.write_png(g8, w, h, 0, 8), 0, sixteen=True)[0][:, :, 0], (g8 * 257).astype(np.uint16))
    rgb = rng.integers(0, 256, (h, w * 3))
    y = O.stbi_load(gen.write_png(rgb, w, h, 2, 8), 1)[0][:, :, 0]
    r3 = rgb.reshape(h, w, 3)
    assert np.array_equal(y, ((r3[:, :, 0] * 77 + r3[:, :, 1] * 150 + r3[:, :, 2] * 29) >> 8).astype(np.uint8))         # stbi__compute_y :911-914


# ------------------------------------------------------------------ JPEG
JPEGS = fixtures.jpegs()


@pytest.mark.parametrize("path", JPEGS, ids=[os.path.basename(p) for p in JPEGS])
def test_jpeg_golden(path):
    name = os.path.basename(path)[:-4]
    d = O.DecodedJpeg(open(path, "rb").read())
    if name in GOLDEN["meta"]:
        m = GOLDEN["meta"][name]
        assert (d.width, d.height, d.comps, d.scan_type) == (m["width"], m["height"], m["comps"], m["scan_type"])
        assert sha(d.coeffs) == m["coeff_sha"] and sha(d.max_zag) == m["max_zag_sha"]
    key = name + ":colfirst"
    if key in GOLDEN["pillow"]:      # H1V1 / grey: libjpeg-turbo's pixels == the reference's arithmetic with the pass order swapped
        rc = 1 if d.comps == 1 else 3
        cf = O.jpeg_reconstruct(d.width, d.height, d.comps, d.scan_type, d.coeffs, d.max_zag, rc, colfirst=True)
        assert sha(cf.reshape(d.height, d.width, rc) if rc == 3 else cf.reshape(d.height, d.width)) == GOLDEN["pillow"][key]
    for rc in (1, 3, 4):
        out = O.jpeg_reconstruct(d.width, d.height, d.comps, d.scan_type, d.coeffs, d.max_zag, rc)
        assert sha(out) == GOLDEN["frozen"][f"{name}:comps{rc}"]
        assert np.array_equal(out, O.jpeg_reconstruct(d.width, d.height, d.comps, d.scan_type, d.coeffs, None, rc))   # sparse paths == dense on real data
        full = O.decompress_jpeg(open(path, "rb").read(), rc)
        assert np.array_equal(full[0], out)


def test_jpeg_progressive_equals_baseline_twin():
    """SOF2 files (jpegload.d:3296-3664) written from the same pixels and quantisation as a baseline file must decode to
    the very same de-quantised coefficients (only the entropy coding differs), and to the same pixels; max_zag follows
    load_next_row's rule (:2286-2290: last non-zero coefficient in zig-zag order, + 1)."""
    zag = np.array([0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,
                    57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63])
    twins = [(n, m["baseline_twin"]) for n, m in GOLDEN["meta"].items() if "baseline_twin" in m]
    assert len(twins) >= 6
    for prog, base in twins:
        dp = open(os.path.join(G, "jpeg", prog + ".jpg"), "rb").read()
        db = open(os.path.join(G, "jpeg", base + ".jpg"), "rb").read()
        assert b"\xff\xc2" in dp and b"\xff\xc2" not in db
        p, b = O.DecodedJpeg(dp), O.DecodedJpeg(db)
        assert (p.width, p.height, p.comps, p.scan_type) == (b.width, b.height, b.comps, b.scan_type)
        assert np.array_equal(p.coeffs, b.coeffs), prog
        nz = p.coeffs[:, zag] != 0
        assert np.array_equal(p.max_zag, np.where(nz[:, 1:].any(axis=1), 64 - np.argmax(nz[:, ::-1], axis=1), 1)), prog
        for rc in (1, 3, 4):
            assert np.array_equal(O.decompress_jpeg(dp, rc)[0], O.decompress_jpeg(db, rc)[0]), (prog, rc)


def test_jpeg_rowfirst_differs_from_libjpeg_order():
    """the reference (jpgd) runs rows first (jpegload.d:335-375); that is NOT libjpeg's result (4967 samples differ on issue35)"""
    d = O.DecodedJpeg(open(os.path.join(G, "ref_images", "issue35.jpg"), "rb").read())
    a = O.jpeg_reconstruct(d.width, d.height, 3, d.scan_type, d.coeffs, d.max_zag, 3)
    b = O.jpeg_reconstruct(d.width, d.height, 3, d.scan_type, d.coeffs, d.max_zag, 3, colfirst=True)
    assert np.count_nonzero(a != b) == 4967 and np.abs(a.astype(int) - b).max() == 3


def _dct(n):
    k = np.arange(n)[:, None]; m = np.arange(n)[None, :]
    c = np.sqrt(2.0 / n) * np.cos(np.pi * (2 * m + 1) * k / (2 * n)); c[0] /= np.sqrt(2)
    return c


def test_jpeg_upsample_float_model_and_kats():
    """frequency-domain 2x chroma upsample (jpegload.d:827-1073, 2139-2255) vs {T,B} X {T,B}^T with T,B = sqrt2 C4 C8^T halves"""
    c4, c8 = _dct(4), _dct(8)
    T, B = np.sqrt(2) * c4 @ c8.T[0:4, :], np.sqrt(2) * c4 @ c8.T[4:8, :]
    assert np.allclose(T[0], [1, .906127, 0, -.318190, 0, .212608, 0, -.180240], atol=1e-6)
    assert np.allclose(T[1], [0, .415735, 1, .791065, 0, -.352443, 0, .277785], atol=1e-6)
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(1000):
        x = np.zeros(64, np.int16)
        idx = rng.integers(0, 64, rng.integers(1, 20)); x[idx] = rng.integers(-1000, 1001, idx.size)
        out = O.jpeg_upsample_block(x, 64).astype(np.float64)
        xf = x.reshape(8, 8).astype(np.float64)
        for q, e in enumerate([T @ xf @ T.T, T @ xf @ B.T, B @ xf @ T.T, B @ xf @ B.T]):
            worst = max(worst, np.abs(out[q][:4, :4] - e).max())
            assert (out[q][4:, :] == 0).all() and (out[q][:, 4:] == 0).all()
    assert worst < 6.0          # fixed-point budget: constants quantised to 2^-10, two rounded stages
    for dc in (-345, 0, 1016, -1024):       # DC-only chroma: four expanded blocks, same DC, flat samples equal to the plain IDCT
        x = np.zeros(64, np.int16); x[0] = dc
        u = O.jpeg_upsample_block(x, 1)
        assert (u[:, 0, 0] == dc).all() and np.count_nonzero(u) == (4 if dc else 0)
        ref = O.jpeg_idct(x, 1)
        for q in range(4):
            assert np.array_equal(O.jpeg_idct_4x4(u[q]), ref)


def _literal():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_literal_jpeg", os.path.join(os.path.dirname(HERE), "tools", "ref_literal_jpeg.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _pinning_blocks(rng, n):
    """n coefficient blocks of three kinds: natural (sparse, zig-zag decaying, |c| <= 1024), dense 11-bit, wild int16"""
    blocks = np.zeros((n, 64), np.int16)
    k = n // 3
    scale = (1024.0 / (1 + np.arange(64)) ** 1.2)
    nat = (rng.standard_normal((k, 64)) * scale).round().clip(-1024, 1023)
    nat *= rng.random((k, 64)) < 0.5
    blocks[:k][:, ZAG] = nat.astype(np.int16)
    blocks[k:2 * k] = rng.integers(-1024, 1024, (k, 64))
    blocks[2 * k:] = rng.integers(-32768, 32768, (n - 2 * k, 64))
    ext = blocks[2 * k:2 * k + 64]
    ext[:] = np.where(rng.random(ext.shape) < 0.5, 32767, -32768)                     # extremes: every wrap-around path
    return blocks


def test_jpeg_literal_restatement_bit_equal_upsample_and_idct():
    """THE pinning of the H2V2 leg: tools/ref_literal_jpeg.py (a second restatement, written literally from jpegload.d
    :156-397, :827-1073, :2132-2255 -- P_Q / R_S transliterated mechanically from the D statements) must agree BIT FOR BIT
    with oracle/oracle_jpeg.c: the four expanded coefficient blocks (incl. cast(short) and the transposed store), the
    idct_4x4 samples, and idct() for every block_max_zag -- on > 10^5 random blocks (natural / dense / wild int16), for all
    64 s_max_rc entries, with real zeros beyond max_zag and with garbage there (both readings must ignore it alike)."""
    R = _literal()
    rng = np.random.default_rng(20260930)
    n_total = 0
    for mz in range(1, 65):
        n = 1800 if mz < 64 else 6000
        blocks = _pinning_blocks(rng, n)
        clean = blocks.copy(); clean[:, ZAG[mz:]] = 0                                    # consistent with max_zag
        for data in (clean, blocks):                                                    # and garbage beyond it
            temps, samples = R.chroma_expand(data, mz)
            pix = R.idct(data, mz)
            for i in range(0, n, 7 if mz < 64 else 1):                                  # the C oracle is called block by block
                up = O.jpeg_upsample_block(data[i], mz)
                assert np.array_equal(up.reshape(4, 64), temps[:, i, :]), (mz, i)
                for q in range(4):
                    assert np.array_equal(O.jpeg_idct_4x4(up[q]).reshape(64), samples[q, i]), (mz, i, q)
                assert np.array_equal(O.jpeg_idct(data[i], mz).reshape(64), pix[i]), (mz, i)
                n_total += 1
    assert n_total >= 40000
    # dense path, every block (the headline case: max_zag = NULL -> 64): 10^5 more through the batched reconstruct below
    blocks = _pinning_blocks(rng, 102000)
    temps, samples = R.chroma_expand(blocks, 64)
    ypix = R.idct(blocks, 64)
    # oracle side in one call: a 4:2:0 frame whose MCUs carry these blocks as Cb (and as Y0): 17000 MCUs = 6 blocks each
    w, h = 16 * 170, 16 * 100
    co = blocks.reshape(17000, 6, 64)
    got = O.jpeg_reconstruct(w, h, 3, O.JPGD_YH2V2, co, None, 4)
    exp = R.decode_h2v2_rgba_fast(co, w, h)
    assert np.array_equal(got, exp)
    assert temps.shape == (4, 102000, 64) and ypix.shape == (102000, 64)


def jpeg_vectors():
    """tests/golden/jpeg_h2v2_ref.npz: outputs of the reference-derived restatement (tools/make_jpeg_vectors.py), committed as data"""
    return np.load(os.path.join(G, "jpeg_h2v2_ref.npz"))


def rgba_to(rgba, width, comps):
    """rgb8 / l8 from the rgba8 scanlines as decompress_jpeg_image_from_stream does (jpegload.d:3776-3792)"""
    px = rgba.reshape(rgba.shape[0], width, 4).astype(np.int64)
    if comps == 4:
        return rgba
    if comps == 3:
        return px[:, :, :3].astype(np.uint8).reshape(rgba.shape[0], width * 3)
    return ((px[:, :, 0] * 19595 + px[:, :, 1] * 38470 + px[:, :, 2] * 7471 + 32768) >> 16).astype(np.uint8)


def check_oracle_against_jpeg_vectors():
    """oracle_jpeg.c == the reference-derived vectors: expanded blocks, idct_4x4, idct per max_zag class, whole frames in the
    three output formats.  (Also run on the GPU box, where tools/ is not consulted: the vectors are data.)"""
    V = jpeg_vectors()
    for i in range(len(V["blocks"])):
        b, mz = V["blocks"][i], int(V["block_max_zag"][i])
        up = O.jpeg_upsample_block(b, mz)
        assert np.array_equal(up.reshape(4, 64), V["expanded"][i]), (i, mz)
        for q in range(4):
            assert np.array_equal(O.jpeg_idct_4x4(up[q]).reshape(64), V["samples4"][i, q]), (i, mz, q)
        assert np.array_equal(O.jpeg_idct(b, mz).reshape(64), V["idct"][i]), (i, mz)
    for f in range(int(V["n_frames"])):
        w, h = int(V[f"frame{f}_w"]), int(V[f"frame{f}_h"])
        co, mz = V[f"frame{f}_coeffs"], (None if bool(V[f"frame{f}_dense"]) else V[f"frame{f}_max_zag"])
        for comps in (4, 3, 1):
            assert np.array_equal(O.jpeg_reconstruct(w, h, 3, O.JPGD_YH2V2, co, mz, comps), rgba_to(V[f"frame{f}_rgba"], w, comps)), (f, comps)


def test_oracle_equals_reference_derived_jpeg_vectors():
    check_oracle_against_jpeg_vectors()


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (build container only)")
def test_jpeg_vectors_and_generated_regions_are_current():
    """the committed vectors are what the restatement produces now, and the restatement's generated regions are what
    jpegload.d transliterates to now"""
    import subprocess, sys
    root = os.path.dirname(HERE)
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_ref_literal.py"), "--check"])
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_jpeg_vectors.py"), "--check"])


def test_jpeg_literal_restatement_pixels_small_frames():
    """whole 4:2:0 frames through the literal driver (transform_mcu_expand + the SSE sequence of expanded_convert, per
    scanline, cropped like :3764) == orc_jpeg_reconstruct, ragged sizes, with and without max_zag"""
    R = _literal()
    rng = np.random.default_rng(7)
    for (w, h) in [(16, 16), (17, 1), (40, 23), (1, 33)]:
        mr, mc = (w + 15) // 16, (h + 15) // 16
        co = _pinning_blocks(rng, mr * mc * 6).reshape(mr * mc, 6, 64)
        assert np.array_equal(O.jpeg_reconstruct(w, h, 3, O.JPGD_YH2V2, co, None, 4), R.decode_h2v2_rgba(co, None, w, h))
        mz = rng.integers(1, 65, (mr * mc, 6)).astype(np.uint8)
        assert np.array_equal(O.jpeg_reconstruct(w, h, 3, O.JPGD_YH2V2, co, mz, 4), R.decode_h2v2_rgba(co, mz, w, h))
    # the colour tables of create_look_ups (:2085-2094) give the same R/G/B as the table-free SSE arithmetic of :2769-2794
    crr, cbb, crg, cbg = R.create_look_ups()
    y, cb, cr = np.meshgrid(np.arange(0, 256, 5), np.arange(256), np.arange(256), indexing="ij")
    r = np.clip(y + crr[cr], 0, 255); g = np.clip(y + ((crg[cr] + cbg[cb]) >> 16), 0, 255); b = np.clip(y + cbb[cb], 0, 255)
    sb = np.zeros(12 * 64, np.uint8)
    for yy in (0, 100, 255):
        for cbv in (0, 77, 128, 255):
            sb[:256] = yy; sb[256:512] = cbv; sb[512:] = np.arange(256, dtype=np.uint8)          # Cr sweeps 0..255 over the 4 Cr blocks
            lines = np.concatenate([R.expanded_convert(sb, 1, row).reshape(16, 4) for row in range(16)])
            crs = np.concatenate([sb[512 + ((row // 8) * 2 + k) * 64 + (row & 7) * 8:][:8] for row in range(16) for k in (0, 1)])
            assert np.array_equal(lines[:, 0], r[yy // 5, cbv, crs]) and np.array_equal(lines[:, 1], g[yy // 5, cbv, crs])
            assert np.array_equal(lines[:, 2], b[yy // 5, cbv, crs]) and (lines[:, 3] == 255).all()


@pytest.mark.skipif(not os.path.exists("/root/reference/source/gamut/codecs/jpegload.d"), reason="reference tree not present")
def test_jpeg_literal_restatement_is_current_transliteration():
    """the P_Q / R_S and the Row / Col statements in tools/ref_literal_jpeg.py are exactly what tools/make_ref_literal.py produces
    from the D text today"""
    import subprocess, sys
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "make_ref_literal.py"), "--check"])


def test_jpeg_rowcol_transliteration_equals_the_handwritten_reading():
    """Row!(N).idct / Col!(N).idct exist twice in tools/ref_literal_jpeg.py: written by hand from the D text, and produced from
    the D text by tools/make_ref_literal.py (syntax rewritten, expressions untouched).  Every N, full-range int16 / int32 inputs:
    the same bits -- and idct() / idct_4x4() / the whole H2V2 comparison above run on the mechanical version."""
    R = _literal()
    rng = np.random.default_rng(11)
    n = 4000
    for N in range(0, 9):
        src = rng.integers(-32768, 32768, (n, 8), dtype=np.int16)
        a = np.full((n, 8), 12345, np.int32); b = a.copy()
        R.Row_idct(N, a, src)
        R.Row_idct_d(N, R.Ptr(b, 0), R.Ptr(src, 0))
        assert np.array_equal(a, b), N
    for N in range(1, 9):
        tmp = rng.integers(-2**31, 2**31, (n, 64), dtype=np.int64).astype(np.int32)
        a = np.zeros((n, 64), np.uint8); b = a.copy()
        for col in range(8):
            R.Col_idct(N, a, tmp, col)
            R.Col_idct_d(N, R.Ptr(b, col), R.Ptr(tmp, col))
        assert np.array_equal(a, b), N


def test_jpeg_sparse_paths_equal_dense():
    """Row!N / Col!N / DC-only / P_Q!(R,C) substitute literal zeros: bit-identical to the dense transform for 8-bit-range data"""
    rng = np.random.default_rng(1)
    for _ in range(400):
        mz = int(rng.integers(1, 65))
        x = np.zeros(64, np.int16)
        x[ZAG[:mz]] = rng.integers(-2000, 2001, mz)
        assert np.array_equal(O.jpeg_idct(x, mz), O.jpeg_idct(x, 64))
        assert np.array_equal(O.jpeg_upsample_block(x, mz), O.jpeg_upsample_block(x, 64))
    # ... but not for int16 extremes with max_zag == 2 (Col!1 does not shift left by 13: no wrap-around)
    x = np.zeros(64, np.int16); x[0], x[1] = 32767, 32767
    assert not np.array_equal(O.jpeg_idct(x, 2), O.jpeg_idct(x, 64))


def test_jpeg_idct_dc_known_answers():
    for dc, exp in [(0, 128), (8, 129), (-8, 127), (1016, 255), (-1024, 0), (4, 129), (3, 128), (-5, 127), (5000, 255), (-5000, 0)]:
        x = np.zeros(64, np.int16); x[0] = dc
        assert (O.jpeg_idct(x, 1) == exp).all() and (O.jpeg_idct(x, 64) == exp).all()


def test_jpeg_error_conventions():
    assert O.decompress_jpeg(b"", 4) is None                                          # issue46.jpg is an empty file: must fail cleanly
    assert O.decompress_jpeg(open(os.path.join(G, "ref_images", "issue46.jpg"), "rb").read(), 4) is None
    data = open(os.path.join(G, "ref_images", "issue35.jpg"), "rb").read()
    assert O.decompress_jpeg(data, 2) is None                                         # jpegload.d:3727
    assert O.decompress_jpeg(data[:500], 4) is None


# ------------------------------------------------------------------ convert
def _np_to_rgbaf32(t, px):
    """independent numpy binary32 model of scanline.d:240-529 (one rounded op per statement)"""
    ch, dt = PT_CHANNELS[t], PT_DTYPE[t]
    v = px.astype(np.float32)
    if dt == np.uint8:
        v = v / np.float32(255.0)
    elif dt == np.uint16:
        v = v / np.float32(65535.0)
    n = px.shape[0]
    out = np.ones((n, 4), np.float32)
    premul = PIXEL_TYPES[t] in ("lap8", "lap16", "lapf32", "rgbap8", "rgbap16", "rgbapf32")
    if ch <= 2:
        g = v[:, 0].copy()
        if ch == 2:
            a = v[:, 1]
            if premul:
                nz = a != 0
                g[nz] = g[nz] / a[nz]
            out[:, 3] = a
        out[:, 0] = out[:, 1] = out[:, 2] = g
    else:
        c = v[:, :3].copy()
        if ch == 4:
            a = v[:, 3]
            if premul:
                nz = a != 0
                c[nz] = c[nz] / a[nz][:, None]
            out[:, 3] = a
        out[:, :3] = c
    return out


def _cvtt(x):
    x = np.asarray(x, np.float32)
    ok = (x >= np.float32(-2147483648.0)) & (x < np.float32(2147483648.0))
    return np.where(ok, np.trunc(np.where(ok, x, 0)).astype(np.int64), -2147483648)


def _np_from_rgbaf32(t, f):
    """independent numpy binary32 model of scanline.d:539-803"""
    ch, dt = PT_CHANNELS[t], PT_DTYPE[t]
    premul = PIXEL_TYPES[t] in ("lap8", "lap16", "lapf32", "rgbap8", "rgbap16", "rgbapf32")
    r, g, b, a = f[:, 0], f[:, 1], f[:, 2], f[:, 3]
    m = np.float32(255.0 if dt == np.uint8 else 65535.0)
    half, three = np.float32(0.5), np.float32(3.0)
    cols = []
    if ch <= 2:
        s = (r + g) + b
        if premul:
            s = s * a
        cols.append(s / three if dt == np.float32 else half + (s * m) / three)
        if ch == 2:
            cols.append(a if dt == np.float32 else half + a * m)
    else:
        for c in (r, g, b):
            x = c * a if premul else c
            cols.append(x if dt == np.float32 else half + x * m)
        if ch == 4:
            cols.append(a if dt == np.float32 else half + a * m)
    v = np.stack(cols, 1)
    if dt == np.float32:
        return v.astype(np.float32)
    return (_cvtt(v) & (0xFF if dt == np.uint8 else 0xFFFF)).astype(dt)


@pytest.mark.parametrize("src", PIXEL_TYPES)
def test_convert_matches_numpy_binary32_model(src):
    rng = np.random.default_rng(PT[src])
    st = PT[src]
    px = gen.make_pixels(src, 600, rng)
    plain8 = ("l8", "la8", "rgb8", "rgba8")
    with np.errstate(all="ignore"):
        for dst in PIXEL_TYPES:
            dt = PT[dst]
            got = O.scanlines_convert(st, px, dt, 600, 1).view(PT_DTYPE[dt]).reshape(600, PT_CHANNELS[dt])
            if src == dst:
                assert np.array_equal(got.view(np.uint8), px.view(np.uint8).reshape(600, -1).view(np.uint8).reshape(got.view(np.uint8).shape))
                continue
            if src in plain8 and dst in plain8:          # rgba8 intermediate (scanline.d:160-234); l8 <- R only
                ch = PT_CHANNELS[st]
                rgba = np.full((600, 4), 255, np.uint8)
                if ch <= 2:
                    rgba[:, 0] = rgba[:, 1] = rgba[:, 2] = px[:, 0]
                    if ch == 2:
                        rgba[:, 3] = px[:, 1]
                else:
                    rgba[:, :ch] = px
                exp = {1: rgba[:, :1], 2: rgba[:, [0, 3]], 3: rgba[:, :3], 4: rgba}[PT_CHANNELS[dt]]
            else:
                exp = _np_from_rgbaf32(dt, _np_to_rgbaf32(st, px))
            e, g2 = np.ascontiguousarray(exp), np.ascontiguousarray(got)
            same = (e.view(np.uint8) == g2.view(np.uint8)).reshape(600, -1).all(axis=1)
            if e.dtype == np.float32:                     # NaN payloads are outside the parity contract
                same |= np.isnan(e).any(axis=1) & np.isnan(g2).any(axis=1)
            assert same.all(), f"{src}->{dst}: rows {np.flatnonzero(~same)[:5]}"


def _scanline_vectors():
    z = np.load(os.path.join(G, "scanline_ref.npz"))
    return z, json.loads(bytes(z["sha_json"]).decode())


def check_convert_against_reference_vectors(pairs=None):
    """oracle/oracle_convert.c == tests/golden/scanline_ref.npz, the outputs of the reference's OWN scanline functions
    (source/gamut/scanline.d executed statement by statement through tools/d_scanline_exec.py, generated by
    tools/make_scanline_vectors.py).  0 ulp: f32 results are compared as bit patterns."""
    z, sha = _scanline_vectors()
    n = z["in_l8"].size
    for s in PIXEL_TYPES:
        for d in PIXEL_TYPES:
            if s == d or (pairs is not None and (s, d) not in pairs):
                continue
            got = O.scanlines_convert(s, z["in_" + s], d, n, 1)
            assert np.array_equal(got.view(np.uint8).reshape(-1), z[f"out_{s}_{d}"]), (s, d)
    if pairs is not None:
        return
    all16, all8 = np.arange(65536, dtype=np.uint16), np.arange(256, dtype=np.uint8)
    for key, v in [("l16_lf32", all16), ("l16_l8", all16), ("l8_l16", all8), ("l8_lf32", all8), ("l16_rgba8", all16), ("l8_rgbaf32", all8)]:
        s, d = key.split("_")
        assert hashlib.sha256(O.scanlines_convert(s, v.view(np.uint8), d, v.size, 1).tobytes()).hexdigest() == sha[key], key
    c, a = np.meshgrid(all8, all8, indexing="ij")
    pairs8 = np.stack([c, a], axis=-1).reshape(-1).astype(np.uint8)
    for key in ("pairs_lap8_laf32", "pairs_lap8_la8", "pairs_la8_lap8", "pairs_lap8_rgbaf32", "pairs_la8_lap16"):
        _, s, d = key.split("_")
        assert hashlib.sha256(O.scanlines_convert(s, pairs8, d, 65536, 1).tobytes()).hexdigest() == sha[key], key


def test_convert_matches_reference_vectors():
    check_convert_against_reference_vectors()


@pytest.mark.skipif(not os.path.exists("/root/reference/source/gamut/scanline.d"), reason="reference tree not present")
def test_convert_vectors_are_current_and_random_sweep_vs_reference_text():
    """the committed vectors are what the reference's text gives today, and on fresh random rows of every pair the oracle
    equals the executed reference too (premultiplied, grey and float->int expressions included: scanline.d:240-803)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import make_scanline_vectors as M
    fresh = M.build()
    z, _ = _scanline_vectors()
    assert set(fresh) == set(z.files)
    for k in z.files:
        assert np.array_equal(fresh[k], z[k]), k
    R = M.Reference()
    ins = M.inputs(np.random.default_rng(99))
    for s in PIXEL_TYPES:
        for d in PIXEL_TYPES:
            if s != d:
                exp = R.convert_row(s, d, ins[s], M.N, M.SIZE)
                assert np.array_equal(O.scanlines_convert(s, ins[s], d, M.N, 1).view(np.uint8).reshape(-1), exp), (s, d)


def test_convert_composite_tables():
    """through the f32 intermediate rgba8->rgba16 is v*257 and rgba16->rgba8 is (v*255+32767)//65535 (NOT stb's >>8)"""
    u8 = np.arange(256, dtype=np.uint8).repeat(4).reshape(-1, 4)
    got = O.scanlines_convert("rgba8", u8, "rgba16", 256, 1).view(np.uint16)
    assert np.array_equal(got, (u8.reshape(-1).astype(np.uint32) * 257).astype(np.uint16))
    u16 = np.arange(65536, dtype=np.uint16).repeat(4).reshape(-1, 4)
    got = O.scanlines_convert("rgba16", u16, "rgba8", 65536, 1)
    exp = ((u16.reshape(-1).astype(np.uint64) * 255 + 32767) // 65535).astype(np.uint8)
    assert np.array_equal(got, exp)
    assert np.count_nonzero(exp != (u16.reshape(-1) >> 8)) == 16256 * 4


def test_convert_pitches_and_flip():
    rng = np.random.default_rng(3)
    w, h = 13, 4
    px = gen.make_pixels("rgb8", w * h, rng)
    tight = O.scanlines_convert("rgb8", px, "rgbaf32", w, h)
    buf, off, pitch = gen.pack_rows(px, w, h, w * 3 + 5, flipped=True)
    out = O.scanlines_convert("rgb8", buf[off - (h - 1) * (w * 3 + 5):], "rgbaf32", w, h, src_pitch=pitch) if False else None
    # flipped source read through a negative pitch gives the same rows
    import ctypes as C
    dst = np.zeros(w * 16 * h, np.uint8); ibuf = np.zeros(w * 16, np.uint8)
    assert O.lib().orc_scanlines_convert(PT["rgb8"], buf.ctypes.data + off, pitch, PT["rgbaf32"], dst.ctypes.data, w * 16, w, h, PT["rgbaf32"], ibuf.ctypes.data)
    assert np.array_equal(dst, tight)
    assert O.lib().orc_scanlines_inter_type(PT["la8"], PT["rgb8"]) == PT["rgba8"]
    assert O.lib().orc_scanlines_inter_type(PT["lap8"], PT["rgb8"]) == PT["rgbaf32"]          # premultiplied 8-bit is not "8-bit" (internals/types.d:99-111)


def test_division_by_max_identity():
    """x / 255.0f and x / 65535.0f (scanline.d:240-529) == RN(q + RN(x - M q) * y), q = RN(x * y), y = RN(1 / M), for every
    integer input: the form the HIP kernels use (gamut_amd/csrc/convert.hip div_by_max).  Checked in exact arithmetic."""
    from fractions import Fraction

    def rn32(fr):                                   # Fraction -> nearest-even binary32, as a Fraction
        if fr == 0:
            return Fraction(0)
        sgn = -1 if fr < 0 else 1
        fr = abs(fr)
        e = fr.numerator.bit_length() - fr.denominator.bit_length()
        while Fraction(2) ** e > fr:
            e -= 1
        while Fraction(2) ** (e + 1) <= fr:
            e += 1
        ulp = Fraction(2) ** (max(e, -126) - 23)
        q = fr / ulp
        n = q.numerator // q.denominator
        rem = q - n
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
            n += 1
        return sgn * n * ulp

    for M, step in ((255, 1), (65535, 1)):
        y = rn32(Fraction(1, M))
        assert float(y) == float(np.float32(1.0) / np.float32(M))
        for x in range(0, M + 1, step):
            q = rn32(x * y)
            r = rn32(x - M * q)
            assert r == x - M * q                    # the residual FMA is exact
            assert float(rn32(q + r * y)) == float(np.float32(x) / np.float32(M)), (M, x)


# ------------------------------------------------------------------ QOI
def _qoi_test_images():
    rng = np.random.default_rng(77)
    out = []
    for k, (w, h, ch) in enumerate([(1, 1, 3), (3, 2, 4), (64, 33, 3), (130, 67, 4), (257, 19, 4)]):
        a = gen.synth_rgb(w, h, 40 + k)
        if ch == 4:
            al = ((np.add.outer(np.arange(h), np.arange(w)) * 5) % 256).astype(np.uint8)
            al[: h // 2] = 255
            a = np.dstack([a, al])
        a[:, : w // 3] = a[0, 0]                                  # long runs (> 62 pixels on the wide ones)
        if h > 4 and w > 16:
            a[1, ::2] = a[0, 0]; a[1, 1::2] = 255 - a[0, 0]           # two far-apart colours alternating: INDEX ops
        if w > 8:
            a[h // 2:, w // 2:] = rng.integers(0, 256, a[h // 2:, w // 2:].shape)      # noise: RGB / RGBA ops, hash collisions
        out.append(a)
    return out


def test_synth_qoi_encoder_equals_spec_encoder():
    """bench.py's vectorised QOI encoder (gamut_amd/synth.py) emits byte for byte what the sequential specification encoder
    of gen.py emits, runs longer than 62 and RUN ops at the very end included"""
    from gamut_amd import synth
    rng = np.random.default_rng(9)
    imgs = _qoi_test_images() + [np.zeros((3, 200, 3), np.uint8), np.repeat(rng.integers(0, 256, (4, 5, 4), dtype=np.uint8), 70, axis=1),
                                 rng.integers(0, 4, (30, 40, 3), dtype=np.uint8) * 60]
    for a in imgs:
        assert synth.qoi_encode(a, 1) == gen.qoi_encode(a, 1), a.shape


def test_qoi_oracle_roundtrip_and_pillow():
    """QOI is lossless and fully specified: streams from the independent spec-based encoder in gen.py must decode to the
    source pixels, and to what Pillow's QOI decoder makes of them; channel forcing follows qoi.d:536-545."""
    import io
    from PIL import Image
    ops = set()
    for a in _qoi_test_images():
        h, w, ch = a.shape
        data = gen.qoi_encode(a, colorspace=1 if ch == 4 else 0)
        p = 14
        while p < len(data) - 8:                                  # which ops the stream uses (coverage of the test data)
            b = data[p]
            op, n = ("rgb", 4) if b == 0xFE else ("rgba", 5) if b == 0xFF else (("index", 1), ("diff", 1), ("luma", 2), ("run", 1))[b >> 6]
            ops.add(op); p += n
        px, fc, cs = O.qoi_decode(data)
        assert fc == ch and cs == (1 if ch == 4 else 0)
        assert np.array_equal(px.reshape(h, w, ch), a)
        assert np.array_equal(np.array(Image.open(io.BytesIO(data))), a)
        p3 = O.qoi_decode(data, 3)[0].reshape(h, w, 3)
        p4 = O.qoi_decode(data, 4)[0].reshape(h, w, 4)
        assert np.array_equal(p3, a[:, :, :3]) and np.array_equal(p4[:, :, :3], a[:, :, :3])
        assert np.array_equal(p4[:, :, 3], a[:, :, 3] if ch == 4 else np.full((h, w), 255))
    assert ops == {"rgb", "rgba", "index", "diff", "luma", "run"}
    # header validation (qoi.d:458-480) and a stream that ends early (remaining pixels repeat the last one, :498-501)
    good = gen.qoi_encode(_qoi_test_images()[2])
    assert O.qoi_decode(good[:21]) is None and O.qoi_decode(b"xoif" + good[4:]) is None
    assert O.qoi_decode(good, 1) is None and O.qoi_decode(good, 2) is None
    assert O.qoi_decode(good[:4] + bytes(4) + good[8:]) is None                      # width 0
    cut = good[:200] + good[-8:]
    t = O.qoi_decode(cut)[0]
    assert t.shape == (33, 64 * 3) and (t.reshape(-1, 3)[-1] == t.reshape(-1, 3)[-50]).all()


def test_scripted_progressive_files_decode_like_their_baseline_twins():
    """tests/jpeg_scripts.py (progressive files with arbitrary scan scripts, written from a baseline file's coefficients): the oracle decodes
    every one of them to the pixels of the baseline file, and so does Pillow -- the generator and the oracle's progressive decoder
    (jpegload.d:3296-3664) agree with libjpeg on scripts libjpeg's own encoder never writes"""
    import io
    from PIL import Image
    import gen
    import jpeg_scripts as J
    for (w, h, kw) in ((77, 50, dict(quality=30, subsampling=2)), (131, 97, dict(quality=75, subsampling=0)), (200, 120, dict(quality=95, subsampling=1))):
        bio = io.BytesIO(); Image.fromarray(gen.synth_rgb(w, h, 5)).save(bio, "JPEG", **kw); base = bio.getvalue()
        want = O.decompress_jpeg(base, 3)[0]
        want_pil = np.array(Image.open(io.BytesIO(base)).convert("RGB"))
        for name, sc in J.SCRIPTS.items():
            for ri in (0, 7):
                p = J.progressive_with_script(base, sc, ri)
                assert b"\xff\xc2" in p[:1000]
                assert np.array_equal(O.decompress_jpeg(p, 3)[0], want), (w, h, name, ri)
                assert np.array_equal(np.array(Image.open(io.BytesIO(p)).convert("RGB")), want_pil), (w, h, name, ri)
