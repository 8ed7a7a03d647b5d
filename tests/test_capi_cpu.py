"""CPU tests of the product side: the C-ABI library loads without a GPU and exports every symbol include/gamut_hip.h
declares; host-side logic (JPEG entropy feeder, PNG header scan, argument validation, error conventions) agrees with
the oracle.  No compute entry point is exercised here (there is no CPU fallback to exercise)."""
import ctypes as C
import fixtures
import os
import re

import numpy as np
import pytest

import gen
import oracle_lib as O
from gamut_amd import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gamut_hip.h")).read()
    declared = set(re.findall(r"\b(gamut_hip_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    raw = C.CDLL(_capi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"libgamut_hip.so does not export {name}"
    assert declared == set(_capi.SIGNATURES), "python binding table and header disagree"
    L = _capi.lib()
    assert b"gfx950" in L.gamut_hip_version()
    # ... and the Image mirror's header
    from gamut_amd import image as gi
    header2 = open(os.path.join(ROOT, "include", "gamut_image.h")).read()
    declared2 = set(re.findall(r"\b(gamut_[a-z0-9_]+)\s*\(", header2))
    for name in sorted(declared2):
        assert hasattr(raw, name), f"libgamut_hip.so does not export {name}"
    assert declared2 == set(gi.IMAGE_SIGNATURES)


def test_enums_mirror_the_reference():
    header = open(os.path.join(ROOT, "include", "gamut_hip.h")).read()
    names = re.search(r"GAMUT_PIXEL_l8 = 0,(.*?)GAMUT_PIXEL_COUNT", header, re.S).group(1)
    order = ["l8"] + re.findall(r"GAMUT_PIXEL_([a-z0-9]+)", names)
    assert order == O.PIXEL_TYPES                                  # types.d:32-59 ordinals
    L = _capi.lib()
    for i, n in enumerate(O.PIXEL_TYPES):
        assert L.gamut_hip_pixel_type_size(i) == O.PT_SIZE[i] == O.lib().orc_pixel_type_size(i)
        for j in range(len(O.PIXEL_TYPES)):
            assert L.gamut_hip_scanlines_inter_type(i, j) == O.lib().orc_scanlines_inter_type(i, j)
    assert L.gamut_hip_pixel_type_size(-1) == 0 and L.gamut_hip_pixel_type_size(18) == 0


JPEGS = fixtures.jpegs()


@pytest.mark.parametrize("path", JPEGS, ids=[os.path.basename(p) for p in JPEGS])
def test_jpeg_feeder_matches_oracle(path):
    """product host feeder (jpeg_host.hip) == oracle feeder: coefficients, max_zag, geometry, JFIF metadata"""
    L = _capi.lib()
    data = open(path, "rb").read()
    buf = np.frombuffer(data, np.uint8)
    fr = _capi.JpegFrame()
    _capi.check(L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, buf.size, C.byref(fr)))
    d = O.DecodedJpeg(data)
    n = fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu
    assert (fr.width, fr.height, fr.comps, fr.scan_type, fr.mcus_per_row, fr.mcus_per_col, fr.blocks_per_mcu) == \
           (d.width, d.height, d.comps, d.scan_type, d.mcus_per_row, d.mcus_per_col, d.blocks_per_mcu)
    assert np.array_equal(np.ctypeslib.as_array(fr.coeffs, (n, 64)), d.coeffs)
    assert np.array_equal(np.ctypeslib.as_array(fr.max_zag, (n,)), d.max_zag)
    assert (fr.pixel_aspect_ratio, fr.dpi_y) == (d.pixel_aspect_ratio, d.dpi_y)
    L.gamut_hip_jpeg_frame_free(C.byref(fr))
    L.gamut_hip_jpeg_frame_free(C.byref(fr))           # idempotent


def _scan_list_files():
    import io
    from PIL import Image
    import gen
    img = gen.synth_rgb(97, 61, 5)
    out = []
    for kw in (dict(quality=85, subsampling=2), dict(quality=85, subsampling=0), dict(quality=85, subsampling=1, restart_marker_blocks=3),
               dict(quality=85, subsampling=2, optimize=True), dict(quality=80, subsampling=2, progressive=True), dict(quality=80, subsampling=0, progressive=True)):
        bio = io.BytesIO(); Image.fromarray(img).save(bio, "JPEG", **kw)
        out.append(bio.getvalue())
    return out


def test_jpeg_feeder_scan_headers_that_list_components_out_of_order():
    """calc_mcu_block_order (jpegload.d:3068-3088) lays the MCU out in the order the SOS lists the components, read_sos_marker (:1466-1540) checks
    neither the order nor repeats: block b is decoded with the tables / predictor of the b-th listed component while the IDCT and the colour
    conversion go by position.  Sequential: a permuted list decodes (chroma planes swapped or garbled), a repeat with the frame's block count too,
    any other repeat has no defined result (the decoder's buffers are sized by init_frame :3136-3260) -- rejected.  Progressive: a repeat in an
    interleaved scan walks out of the component's plane (decode_scan :3520-3583, coeff_buf_getp's assert :3293) -- rejected.  Host feeder == oracle."""
    import gen
    L = _capi.lib()
    n_ok = n_rej = n_changed = 0
    for f in _scan_list_files():
        base = O.DecodedJpeg(f)
        for v in gen.sos_component_lists(f):
            try:
                d = O.DecodedJpeg(v)
            except ValueError:
                d = None
            fr = _capi.JpegFrame(); buf = np.frombuffer(v, np.uint8)
            rc = L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, buf.size, C.byref(fr))
            assert (rc == 0) == (d is not None), (rc, L.gamut_hip_last_error())
            if d is None:
                assert rc == _capi.ERR_DECODE
                n_rej += 1
                continue
            n = fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu
            assert np.array_equal(np.ctypeslib.as_array(fr.coeffs, (n, 64)), d.coeffs) and np.array_equal(np.ctypeslib.as_array(fr.max_zag, (n,)), d.max_zag)
            L.gamut_hip_jpeg_frame_free(C.byref(fr))
            n_ok += 1; n_changed += not np.array_equal(d.coeffs, base.coeffs)
    assert n_ok >= 20 and n_rej >= 10 and n_changed >= 15, (n_ok, n_rej, n_changed)


FUZZ_FOUND = sorted(os.path.join(G, "jpeg_fuzz", n) for n in os.listdir(os.path.join(G, "jpeg_fuzz")) if n.endswith(".jpg"))


@pytest.mark.parametrize("path", FUZZ_FOUND, ids=[os.path.basename(p) for p in FUZZ_FOUND])
def test_jpeg_feeder_on_the_files_the_fuzzers_found(path):
    """tests/golden/jpeg_fuzz: the files on which a GPU decoder and the oracle once disagreed (tools/fuzz_prog_gpu.py, tools/fuzz_mixed_gpu.py; what each
    one showed is told at its GPU test / in DESIGN.md).  Host feeder == oracle: verdict, and coefficients where both decode."""
    L = _capi.lib()
    data = open(path, "rb").read()
    try:
        d = O.DecodedJpeg(data)
    except ValueError:
        d = None
    fr = _capi.JpegFrame(); buf = np.frombuffer(data, np.uint8)
    rc = L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, buf.size, C.byref(fr))
    assert (rc == 0) == (d is not None), (rc, L.gamut_hip_last_error())
    if d is not None:
        n = fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu
        assert np.array_equal(np.ctypeslib.as_array(fr.coeffs, (n, 64)), d.coeffs) and np.array_equal(np.ctypeslib.as_array(fr.max_zag, (n,)), d.max_zag)
        L.gamut_hip_jpeg_frame_free(C.byref(fr))


def test_jpeg_feeder_octets_between_a_restart_interval_and_its_marker():
    """process_restart (jpegload.d:2335-2402) to the letter, on the CPU: the host feeder's resync() against the oracle's restart() on files with
    octets between an interval's last bit and its RSTn -- skipped if none is 0xFF and the marker lies within the 1536 bytes the function reads from
    where the decoder's input stands, JPGD_BAD_RESTART_MARKER otherwise (tests/test_jpeg_gpu.py runs the same files through the device decoders).
    Baseline (markers early and late in the file) and progressive (a marker of every scan kind)."""
    import io
    from PIL import Image
    import gen
    import jpeg_scripts as JS
    L = _capi.lib()
    bio = io.BytesIO(); Image.fromarray(gen.synth_rgb(160, 96, 32)).save(bio, "JPEG", quality=90, subsampling=2, restart_marker_blocks=3); small = bio.getvalue()
    prog = JS.progressive_with_script(small, JS.LIBJPEG_DEFAULT, restart=4)
    n_prog = len(gen.rst_positions(prog))
    blobs = gen.leftover_variants(small, 2) + gen.leftover_variants(small, len(gen.rst_positions(small)) - 1)
    for which in range(0, n_prog, max(1, n_prog // 12)):
        blobs += gen.leftover_variants(prog, which)
    n_ok = n_rej = 0
    for k, b in enumerate(blobs):
        try:
            d = O.DecodedJpeg(b)
        except ValueError:
            d = None
        fr = _capi.JpegFrame(); buf = np.frombuffer(b, np.uint8)
        rc = L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, buf.size, C.byref(fr))
        assert (rc == 0) == (d is not None), (k, rc, L.gamut_hip_last_error())
        if d is None:
            n_rej += 1
            continue
        n = fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu
        assert np.array_equal(np.ctypeslib.as_array(fr.coeffs, (n, 64)), d.coeffs) and np.array_equal(np.ctypeslib.as_array(fr.max_zag, (n,)), d.max_zag), k
        L.gamut_hip_jpeg_frame_free(C.byref(fr))
        n_ok += 1
    assert n_ok > len(blobs) // 4 and n_rej > len(blobs) // 4, (n_ok, n_rej)


def test_jpeg_batch_feeder_threads():
    """gamut_hip_jpeg_decode_coeffs_batch: N independent files on a thread pool == one at a time; a bad file fails alone."""
    import time
    L = _capi.lib()
    blobs = [open(p, "rb").read() for p in JPEGS] * 6 + [b"\xff\xd8garbage"]
    bufs = [np.frombuffer(b, np.uint8) for b in blobs]
    n = len(blobs)
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    lens = (C.c_size_t * n)(*[b.size for b in bufs])
    for threads in (1, 4, 0):
        frames = (_capi.JpegFrame * n)()
        status = (C.c_int * n)()
        rc = L.gamut_hip_jpeg_decode_coeffs_batch(ptrs, lens, n, frames, status, threads)
        assert rc == _capi.ERR_DECODE and L.gamut_hip_last_error().startswith(b"image %d:" % (n - 1))
        assert list(status[:n - 1]) == [0] * (n - 1) and status[n - 1] == _capi.ERR_DECODE and not frames[n - 1].coeffs
        for i in range(n - 1):
            d = O.DecodedJpeg(blobs[i])
            k = frames[i].mcus_per_row * frames[i].mcus_per_col * frames[i].blocks_per_mcu
            assert np.array_equal(np.ctypeslib.as_array(frames[i].coeffs, (k, 64)), d.coeffs)
            assert np.array_equal(np.ctypeslib.as_array(frames[i].max_zag, (k,)), d.max_zag)
            L.gamut_hip_jpeg_frame_free(C.byref(frames[i]))
    frames = (_capi.JpegFrame * n)()
    assert L.gamut_hip_jpeg_decode_coeffs_batch(ptrs, lens, n - 1, frames, None, 3) == 0          # all good, no status array
    for i in range(n - 1):
        L.gamut_hip_jpeg_frame_free(C.byref(frames[i]))
    assert L.gamut_hip_jpeg_decode_coeffs_batch(None, None, 0, None, None, 2) == 0
    assert L.gamut_hip_jpeg_decode_coeffs_batch(None, None, 3, None, None, 2) == _capi.ERR_INVALID_ARG


def test_plain_c99_consumer(tmp_path):
    """the headers are C (not C++): a strict C99 program includes both, links the library and uses it"""
    import subprocess
    exe = str(tmp_path / "abi_consumer")
    lib_dir = os.path.dirname(_capi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_consumer.c"), "-o", exe, "-L", lib_dir, "-lgamut_hip", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([exe, os.path.join(G, "jpeg", "cfg1_640x480_420_q90.jpg")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.startswith("640x480 comps=3 scan_type=4 blocks=7200 ") and out.stdout.rstrip().endswith("format=0")


def test_jpeg_scan_layout():
    """segments the device entropy decoder would get: 1 without restart markers, one per interval with them"""
    L = _capi.lib()
    for name, nseg in (("s_131x97_420.jpg", 1), ("s_131x97_420_rst.jpg", None), ("p_131x97_420.jpg", -1)):
        buf = np.frombuffer(open(os.path.join(G, "jpeg", name), "rb").read(), np.uint8)
        fr = _capi.JpegFrame(); seg = C.c_int32(); nbytes = C.c_uint64()
        rc = L.gamut_hip_jpeg_scan_layout(buf.ctypes.data, buf.size, C.byref(fr), C.byref(seg), C.byref(nbytes))
        if nseg == -1:                                                         # progressive: the segments of all its scans (jpeg_prog.hpp)
            assert rc == 0 and (fr.width, fr.height) == (131, 97)
            assert seg.value == buf.tobytes().count(b"\xff\xda") and 0 < nbytes.value <= buf.size + 84 * seg.value
            continue
        assert rc == 0 and (fr.width, fr.height) == (131, 97)
        mcus = fr.mcus_per_row * fr.mcus_per_col
        assert seg.value == (1 if nseg == 1 else (mcus + 2) // 3)             # restart_marker_blocks=3 in tools/make_fixtures.py
        assert 0 < nbytes.value <= buf.size + 64 * seg.value


def test_jpeg_read_header_needs_no_device():
    """gamut_hip_jpeg_read_header: geometry without entropy decoding (what a caller sizes its device buffers from)"""
    L = _capi.lib()
    for path in JPEGS:
        data = open(path, "rb").read()
        buf = np.frombuffer(data, np.uint8)
        fr = _capi.JpegFrame()
        _capi.check(L.gamut_hip_jpeg_read_header(buf.ctypes.data, buf.size, C.byref(fr)))
        d = O.DecodedJpeg(data)
        assert (fr.width, fr.height, fr.comps, fr.scan_type, fr.mcus_per_row, fr.mcus_per_col, fr.blocks_per_mcu) == \
               (d.width, d.height, d.comps, d.scan_type, d.mcus_per_row, d.mcus_per_col, d.blocks_per_mcu)
        assert not fr.coeffs and not fr.max_zag and (fr.pixel_aspect_ratio, fr.dpi_y) == (d.pixel_aspect_ratio, d.dpi_y)
    fr = _capi.JpegFrame()
    assert L.gamut_hip_jpeg_read_header(None, 0, C.byref(fr)) == _capi.ERR_DECODE
    assert L.gamut_hip_jpeg_read_header(None, 0, None) == _capi.ERR_INVALID_ARG


def test_qoi_header_needs_no_device():
    L = _capi.lib()
    a = gen.synth_rgb(37, 11, 2)
    data = np.frombuffer(gen.qoi_encode(np.dstack([a, a[:, :, 0]]), colorspace=1), np.uint8)
    d = _capi.QoiDesc()
    _capi.check(L.gamut_hip_qoi_read_header(data.ctypes.data, data.size, C.byref(d)))
    assert (d.width, d.height, d.channels, d.colorspace) == (37, 11, 4, 1)
    bad = data.copy(); bad[0] = 0x78
    assert L.gamut_hip_qoi_read_header(bad.ctypes.data, bad.size, C.byref(d)) == _capi.ERR_DECODE
    assert L.gamut_hip_qoi_read_header(data.ctypes.data, 21, C.byref(d)) == _capi.ERR_DECODE
    assert L.gamut_hip_qoi_read_header(data.ctypes.data, data.size, None) == _capi.ERR_INVALID_ARG


def test_jpeg_rejects_huffman_tables_that_are_not_prefix_codes():
    """a DHT whose length counts over-subscribe the code space (found by tools/fuzz_host.sh: it overflowed the look-up table)"""
    L = _capi.lib()
    good = open(os.path.join(G, "jpeg", "s_16x16_420.jpg"), "rb").read()
    i = good.index(b"\xff\xc4")
    bad = bytearray(good)
    counts = list(bad[i + 5:i + 21])
    k = max(range(16), key=lambda j: counts[j])
    assert counts[k] >= 3
    bad[i + 5] = counts[0] + 3; bad[i + 5 + k] = counts[k] - 3          # three codes of length 1, same number of symbols
    fr = _capi.JpegFrame()
    buf = np.frombuffer(bytes(bad), np.uint8)
    assert L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, buf.size, C.byref(fr)) == _capi.ERR_DECODE
    assert b"prefix code" in L.gamut_hip_last_error() and not fr.coeffs
    assert L.gamut_hip_jpeg_read_header(buf.ctypes.data, buf.size, C.byref(fr)) == _capi.ERR_DECODE
    with pytest.raises(ValueError):
        O.DecodedJpeg(bytes(bad))


def test_jpeg_feeder_rejects_bad_streams():
    L = _capi.lib()
    fr = _capi.JpegFrame()
    good = open(os.path.join(G, "ref_images", "issue35.jpg"), "rb").read()
    cases = [b"", b"\xff\xd8", b"\x89PNG\r\n\x1a\n" + b"0" * 64, good[:200],
             good.replace(b"\xff\xc0", b"\xff\xc2", 1)]                                  # baseline scan under a progressive SOF: bad spectral selection
    for data in cases:
        buf = np.frombuffer(data, np.uint8) if data else np.zeros(1, np.uint8)
        rc = L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, len(data), C.byref(fr))
        assert rc == _capi.ERR_DECODE and L.gamut_hip_last_error() != b"", data[:8]
        assert not fr.coeffs and not fr.max_zag
        with pytest.raises(ValueError):
            O.DecodedJpeg(data)
    # a file that ends inside its scan is NOT refused: jpgd pads the stream with FF D9 (get_char, jpegload.d:631-652), the bit reader hands out 1-bits at
    # the marker (get_octet :683-696) and a bit pattern no code word begins decodes as symbol 0 (huff_decode :746-813) -- the rest of the picture is flat
    cut = good[:4000]
    buf = np.frombuffer(cut, np.uint8)
    assert L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, buf.size, C.byref(fr)) == _capi.OK
    d = O.DecodedJpeg(cut)
    n = fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu
    assert np.array_equal(np.ctypeslib.as_array(fr.coeffs, (n, 64)), d.coeffs) and np.array_equal(np.ctypeslib.as_array(fr.max_zag, (n,)), d.max_zag)
    assert not d.coeffs[-1].any() or d.max_zag[-1] == 1
    L.gamut_hip_jpeg_frame_free(C.byref(fr))
    assert L.gamut_hip_jpeg_decode_coeffs(None, 0, None) == _capi.ERR_INVALID_ARG


def test_png_is16():
    L = _capi.lib()
    for name, exp in [("issue76.png", 1), ("issue65.png", 0), ("vst3-compatible.png", 0), ("issue35.jpg", 0)]:
        buf = np.frombuffer(open(os.path.join(G, "ref_images", name), "rb").read(), np.uint8)
        assert L.gamut_hip_png_is16(buf.ctypes.data, buf.size) == exp


def test_no_gpu_means_loud_failure_not_fallback():
    """on a box without a GPU every compute entry point reports an error; nothing is computed on the CPU"""
    L = _capi.lib()
    if L.gamut_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    a = np.zeros(16, np.uint8); b = np.full(64, 0xA5, np.uint8)
    rc = L.gamut_hip_scanlines_convert(O.PT["rgba8"], a.ctypes.data, 16, O.PT["rgbaf32"], b.ctypes.data, 64, 4, 1)
    assert rc == _capi.ERR_NO_DEVICE and b"no HIP device" in L.gamut_hip_last_error()
    assert (b == 0xA5).all()
    assert L.gamut_hip_init(0) == _capi.ERR_NO_DEVICE
    with pytest.raises(_capi.GamutHipError):
        _capi.check(rc)
    buf = np.frombuffer(open(os.path.join(G, "ref_images", "issue35.jpg"), "rb").read(), np.uint8)
    w, h, ac = C.c_int(), C.c_int(), C.c_int()
    par, dpi = C.c_float(), C.c_float()
    assert not L.gamut_hip_decompress_jpeg_image_from_memory(buf.ctypes.data, buf.size, C.byref(w), C.byref(h), C.byref(ac), C.byref(par), C.byref(dpi), 4)
    png = np.frombuffer(open(os.path.join(G, "ref_images", "issue76.png"), "rb").read(), np.uint8)
    f = C.c_float()
    assert not L.gamut_hip_stbi_load_from_memory(png.ctypes.data, png.size, C.byref(w), C.byref(h), C.byref(ac), 0, C.byref(f), C.byref(f), C.byref(f))
    assert b"no HIP device" in L.gamut_hip_last_error()
    ptrs = (C.c_void_p * 1)(buf.ctypes.data); lens = (C.c_size_t * 1)(buf.size); off = (C.c_int64 * 1)(0)
    info = (_capi.JpegFrame * 1)()
    assert L.gamut_hip_jpeg_entropy_decode_device(ptrs, lens, 1, off, off, 0x1000, 0x1000, None, info, None, None) == _capi.ERR_NO_DEVICE
    qoi = np.frombuffer(gen.qoi_encode(gen.synth_rgb(5, 4, 1)), np.uint8)
    qd = _capi.QoiDesc()
    assert not L.gamut_hip_qoi_decode(qoi.ctypes.data, qoi.size, C.byref(qd), 0) and b"no HIP device" in L.gamut_hip_last_error()
    assert (qd.width, qd.height, qd.channels) == (5, 4, 3)                 # the header was read; nothing was decoded on the CPU


def test_argument_validation_needs_no_device():
    L = _capi.lib()
    d = np.zeros(64, np.uint8)
    p = d.ctypes.data
    assert L.gamut_hip_scanlines_convert_device(-1, p, 4, 0, 12, p, 4, 0, 1, 1, 1, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_scanlines_convert_device(12, p, 4, 0, 18, p, 4, 0, 1, 1, 1, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_scanlines_convert_device(12, p, 4, 0, 14, p, 16, 0, 0, 5, 1, None) == _capi.OK          # zero-size: nothing to do (image.d:1217-1224)
    assert L.gamut_hip_jpeg_reconstruct_batch_device(p, 0, None, 0, p, 4, 0, 0, 8, 4, 4, 1, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_jpeg_reconstruct_batch_device(p, 0, None, 0, p, 4, 0, 20000, 8, 4, 4, 1, None) == _capi.ERR_INVALID_ARG    # > 16384 (jpegload.d:102)
    assert L.gamut_hip_jpeg_reconstruct_batch_device(p, 0, None, 0, p, 4, 0, 8, 8, 7, 4, 1, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_jpeg_reconstruct_batch_device(p, 0, None, 0, p, 4, 0, 8, 8, 4, 2, 1, None) == _capi.ERR_INVALID_ARG        # 2 comps never produced by the codec (jpeg.d:55-56)
    assert L.gamut_hip_png_defilter_batch_device(p, 0, 64, p, 0, 4, 4, 4, 4, 3, 6, 1, None, None) == _capi.ERR_INVALID_ARG       # depth 3
    assert L.gamut_hip_png_defilter_batch_device(p, 0, 64, p, 0, 4, 4, 4, 2, 8, 6, 1, None, None) == _capi.ERR_INVALID_ARG       # out_n < img_n
    assert L.gamut_hip_png_defilter_batch_device(p, 0, 10, p, 0, 4, 4, 4, 4, 8, 6, 1, None, None) == _capi.ERR_DECODE            # not enough pixels (stbdec.d:1430)
    assert L.gamut_hip_png_defilter_batch_device(p, 0, 64, p, 0, 0, 4, 4, 4, 8, 6, 1, None, None) == _capi.ERR_INVALID_ARG       # 0-pixel image (stbdec.d:1897)
    assert b"png_defilter" in L.gamut_hip_last_error()
    # the file-level batch entry points: bad counts / null arrays are argument errors before anything else; empty batches are fine
    assert L.gamut_hip_jpeg_entropy_decode_device(None, None, -1, None, None, None, None, None, None, None, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_jpeg_entropy_decode_device(None, None, 2, None, None, None, None, None, None, None, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_jpeg_entropy_decode_device(None, None, 0, None, None, None, None, None, None, None, None) == _capi.OK
    assert L.gamut_hip_png_decode_batch_device(None, None, 1, 0, 8, None, None, None, None, 1, None) == _capi.ERR_INVALID_ARG
    ptrs = (C.c_void_p * 1)(p); lens = (C.c_size_t * 1)(64); off = (C.c_int64 * 1)(0); info = (_capi.PngInfo * 1)()
    assert L.gamut_hip_png_decode_batch_device(ptrs, lens, 1, 5, 8, off, p, info, None, 1, None) == _capi.ERR_INVALID_ARG        # req_comp 5
    assert L.gamut_hip_png_decode_batch_device(ptrs, lens, 1, 0, 12, off, p, info, None, 1, None) == _capi.ERR_INVALID_ARG       # 12-bit output
    assert L.gamut_hip_png_decode_batch_device(None, None, 0, 0, 8, None, None, None, None, 1, None) == _capi.OK
    assert L.gamut_hip_qoi_decode_batch_device(None, None, 3, 4, None, None, None, None, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_qoi_decode_batch_device(None, None, 0, 4, None, None, None, None, None) == _capi.OK
    assert L.gamut_hip_qoi_decode_resident_device(None, 0, None, None, None, 2, 4, None, None, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_qoi_decode_resident_device(None, 0, None, None, None, 0, 7, None, None, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_qoi_decode_resident_device(None, 0, None, None, None, 0, 4, None, None, None) == _capi.OK
    assert L.gamut_hip_jpeg_decode_coeffs_batch(None, None, 0, None, None, 4) == _capi.OK
    hd = _capi.PngInfo()
    assert L.gamut_hip_png_read_header(p, 64, C.byref(hd)) == _capi.ERR_DECODE and L.gamut_hip_png_read_header(p, 64, None) == _capi.ERR_INVALID_ARG


def _strip_comments(text):
    import re
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))


def _split_args(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        depth += ch == "("
        depth -= ch == ")"
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip() not in ("", "void"):
        out.append(cur)
    return out


_C_SCALARS = {"int": "int", "int32_t": "int", "uint32_t": "uint", "unsigned": "uint", "int64_t": "long", "uint64_t": "ulong", "unsigned long long": "ulong",
              "int16_t": "short", "uint16_t": "ushort", "uint8_t": "ubyte", "int8_t": "byte", "unsigned char": "ubyte", "char": "char", "float": "float",
              "double": "double", "size_t": "size_t", "void": "void"}


def _c_type_to_d(t):
    """`const uint8_t* const*` -> `const(ubyte*)*`: the D spelling of a C parameter type (names already stripped)."""
    import re
    t = re.sub(r"\s+", " ", t.replace("*", " * ")).strip()
    toks = t.split(" ")
    base, i = [], 0
    lead_const = False
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            lead_const = True
        elif toks[i] != "struct":
            base.append(toks[i])
        i += 1
    b = " ".join(base)
    d = _C_SCALARS.get(b, b)                                  # gamut_hip_* struct and function-pointer typedef names stay
    cur, const_inner = d, lead_const                           # const_inner: the thing `cur` names is const
    while i < len(toks):
        assert toks[i] == "*", t
        i += 1
        ptr_const = i < len(toks) and toks[i] == "const"
        if ptr_const:
            i += 1
        cur = (f"const({cur})*" if const_inner else f"{cur}*")
        const_inner = ptr_const
    assert not const_inner or cur == d, t                      # a top-level const on a by-value parameter does not occur
    import re as _re
    while True:                                                # D's const is transitive: const(const(T)*) is spelled const(T*)
        nxt = _re.sub(r"const\(const\(([^()]*)\)(\**)\)", r"const(\1\2)", cur)
        if nxt == cur:
            return cur
        cur = nxt


def _c_param_type(p):
    """drops the parameter's name (the last identifier when the declaration has more than a type)"""
    import re
    p = re.sub(r"\s+", " ", p).strip()
    m = re.match(r"^(.*?[\s\*])([A-Za-z_]\w*)$", p)
    if m and m.group(1).strip() and m.group(1).strip() not in ("const", "unsigned", "struct", "unsigned long"):
        return m.group(1).strip()
    return p


def _d_param_type(p):
    import re
    p = re.sub(r"\s+", " ", p).strip()
    m = re.match(r"^(.*[\s\*\)])([A-Za-z_]\w*)$", p)
    return re.sub(r"\s+", "", (m.group(1) if m else p).strip()) if True else p


def test_d_binding_lists_every_export():
    """bindings/gamut_hip.d (the file INTEGRATION.md tells a maintainer to add) declares every function of include/gamut_hip.h with the
    same RETURN and PARAMETER TYPES -- each C type put through a C -> D type map (`const uint8_t* const*` -> `const(ubyte*)*`,
    `int64_t` -> `long`, ...) and compared with what the D file says, position by position: a wrong width, a missing `const`, a swapped
    pair of arguments of different types fail here (round 4 compared parameter COUNTS).  Inside an extern(C) block, with the four
    extern(C) trampolines for the reference's extern(D) callbacks (plugins/jpeg.d:167, stbdec.d:143-165).  No D compiler exists in
    the image: a textual check."""
    import re
    hdr = _strip_comments(open(os.path.join(ROOT, "include", "gamut_hip.h")).read())
    hdr = re.sub(r"^\s*#[^\n]*", "", hdr, flags=re.M)                   # preprocessor lines
    dsrc = _strip_comments(open(os.path.join(ROOT, "bindings", "gamut_hip.d")).read())

    def protos(text, lang):
        out = {}
        for m in re.finditer(r"([A-Za-z_][\w\s\*\(\)]*?)\b(gamut_hip_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
            ret = re.sub(r"\s+", " ", m.group(1)).strip()
            if "typedef" in ret or ret.endswith("(") or ret.startswith("alias") or "=" in ret:
                continue
            ret = re.sub(r"^(extern \"C\" |GAMUT_HIP_API )", "", ret)
            args = _split_args(m.group(3).strip())
            if lang == "c":
                out[m.group(2)] = (re.sub(r"\s+", "", _c_type_to_d(ret)), [re.sub(r"\s+", "", _c_type_to_d(_c_param_type(a))) for a in args])
            else:
                out[m.group(2)] = (re.sub(r"\s+", "", ret), [_d_param_type(a) for a in args])
        return out
    h, d = protos(hdr, "c"), protos(dsrc, "d")
    assert len(h) >= 60, len(h)
    missing = sorted(set(h) - set(d))
    assert not missing, missing
    for name in sorted(h):
        assert d[name] == h[name], (name, "header (as D):", h[name], "binding:", d[name])
    # the one function-pointer typedef: int (*)(void*, int, unsigned char*, void*)  ==  the alias's int function(void*, int, bool*, void*)
    m = re.search(r"typedef\s+int\s*\(\*gamut_hip_jpeg_stream_read_func\)\s*\(([^)]*)\)", hdr)
    cb = [re.sub(r"\s+", "", _c_type_to_d(_c_param_type(a))) for a in _split_args(m.group(1))]
    m = re.search(r"alias\s+gamut_hip_jpeg_stream_read_func\s*=\s*int\s+function\(([^)]*)\)", dsrc)
    assert [x.replace("bool*", "ubyte*") for x in (_d_param_type(a) for a in _split_args(m.group(1)))] == cb
    assert dsrc.index("extern(C)") < dsrc.index("gamut_hip_version")
    for name in ("gamut_hip_tramp_read_jpeg", "gamut_hip_tramp_stb_read", "gamut_hip_tramp_stb_skip", "gamut_hip_tramp_stb_eof"):
        assert re.search(r"extern\(C\)\s+\w+\s+" + name, dsrc), name


def test_d_binding_struct_layouts(tmp_path):
    """Every struct of include/gamut_hip.h three ways: (1) what the C compiler lays out (tests/c/abi_layout.c: sizeof and every offsetof),
    (2) the `static assert`s of bindings/gamut_hip.d -- what a D compiler will check on the binding's first build --, (3) the layout the D
    declarations of that file yield under the C ABI rules extern(C) D structs follow (natural alignment, declaration order), computed
    here from the D text with D's type sizes.  All three must say the same: a field of the wrong width, a swapped pair, a missing
    member in the binding moves an offset and fails."""
    import re
    import subprocess
    exe = str(tmp_path / "abi_layout")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "c", "abi_layout.c"), "-o", exe])
    c_layout = {}
    for line in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines():
        f = line.split()
        if f:
            c_layout[f[0]] = (int(f[1]), {kv.split("=")[0]: int(kv.split("=")[1]) for kv in f[2:]})
    hdr = _strip_comments(open(os.path.join(ROOT, "include", "gamut_hip.h")).read())
    declared = set(re.findall(r"typedef struct (gamut_hip_\w+)\s*\{", hdr))
    assert declared == set(c_layout), (declared, set(c_layout))           # abi_layout.c knows every struct with a body
    dsrc = _strip_comments(open(os.path.join(ROOT, "bindings", "gamut_hip.d")).read())
    # (2) the static asserts
    asserted = {}
    for m in re.finditer(r"static assert\((.*?)\);", dsrc, flags=re.S):
        for name, field, kind, val in re.findall(r"(gamut_hip_\w+)\.(?:(\w+)\.)?(sizeof|offsetof) == (\d+)", m.group(1)):
            e = asserted.setdefault(name, [None, {}])
            if kind == "sizeof":
                e[0] = int(val)
            else:
                e[1]["out" if field == "out_" else field] = int(val)
    assert {k: (v[0], v[1]) for k, v in asserted.items()} == c_layout
    # (3) the D declarations themselves
    size = {"byte": 1, "ubyte": 1, "bool": 1, "char": 1, "short": 2, "ushort": 2, "int": 4, "uint": 4, "float": 4, "long": 8, "ulong": 8, "size_t": 8, "double": 8}
    for name, (c_size, c_fields) in c_layout.items():
        m = re.search(r"struct " + name + r"\s*\{(.*?)\}", dsrc, flags=re.S)
        assert m, name
        off, align, fields = 0, 1, {}
        for decl in [x.strip() for x in m.group(1).split(";") if x.strip()]:
            fm = re.match(r"^(\w+\s+function\([^)]*\))\s+(\w+)$", decl)
            if fm:
                typ, names = "ptr", [fm.group(2)]
            else:
                tm = re.match(r"^(.*?[\s\*\)])([\w\s,]+)$", decl)
                typ, names = tm.group(1).strip(), [n.strip() for n in tm.group(2).split(",")]
            sz = 8 if typ == "ptr" or typ.endswith("*") else size[typ]
            for n in names:
                off = (off + sz - 1) // sz * sz
                fields["out" if n == "out_" else n] = off
                off += sz
                align = max(align, sz)
        total = (off + align - 1) // align * align
        assert (total, fields) == (c_size, c_fields), (name, "D declaration:", total, fields, "C:", c_size, c_fields)


def test_identify_format_and_mixed_batch_arguments():
    """gamut_hip_identify_format = the plugins' signature tests (plugins/jpeg.d:106-110, png.d:165-169, qoi.d:143-147); the mixed batch call
    validates its arguments and, without a GPU, fails loudly"""
    L = _capi.lib()
    cases = [(b"\xff\xd8\xff\xe0", 0), (b"\x89PNG\r\n\x1a\n....", 1), (b"qoif\0\0", 2), (b"\x89PNG\r\n", -1), (b"\xff", -1), (b"", -1), (b"GIF89a", -1)]
    for data, want in cases:
        buf = np.frombuffer(data + b"\0", np.uint8)
        assert L.gamut_hip_identify_format(buf.ctypes.data, len(data)) == want, data
    assert L.gamut_hip_identify_format(None, 10) == -1
    assert L.gamut_hip_decode_batch_device(None, None, 0, 4, None, None, None, None, None) == 0
    assert L.gamut_hip_decode_batch_device(None, None, 2, 4, None, None, None, None, None) == _capi.ERR_INVALID_ARG
    buf = np.frombuffer(b"qoif" + bytes(20), np.uint8)
    ptrs = (C.c_void_p * 1)(buf.ctypes.data); lens = (C.c_size_t * 1)(buf.size); offs = (C.c_int64 * 1)(0); info = (_capi.ImageInfo * 1)()
    assert L.gamut_hip_decode_batch_device(ptrs, lens, 1, 2, offs, 1, info, None, None) == _capi.ERR_INVALID_ARG      # req_comps 3 or 4
    if L.gamut_hip_device_count() == 0:
        assert L.gamut_hip_decode_batch_device(ptrs, lens, 1, 4, offs, 1, info, None, None) == _capi.ERR_NO_DEVICE
