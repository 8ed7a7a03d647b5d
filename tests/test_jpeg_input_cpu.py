"""The JPEG input layer on the CPU: the oracle (oracle/oracle_jpeg_input.c) and the product's host feeder (gamut_hip_jpeg_decode_coeffs) against
tests/golden/jpeg_fuzz/expected.json -- what the SECOND READING of jpegload.d (tools/ref_literal_input.py, build-container only; the file is
written by tools/make_jpeg_fuzz_fixtures.py) says about every file of that directory: verdict, geometry, pixelAspectRatio / dotsPerInchY,
SHA-256 of every coefficient and every max_zag.  Each file stands on one habit of jpgd's input layer (its name says which; the reference lines
are cited in the generator): APP1 / EXIF density and its three reject cases, NaN where no segment carries a density, the FF D9 padding behind
the end of the stream, symbol 0 for a bit pattern no code word begins, find_eoi's walk behind the last MCU row, RSTn / TEM / JPG / SOI / SOF3
between segments, locate_soi_marker's 4097 bytes, table selectors, restart structures.  -m gpu: tests/test_jpeg_gpu.py runs the same files
through the device decoders."""
import ctypes as C
import hashlib
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O
from gamut_amd import _capi

HERE = os.path.dirname(os.path.abspath(__file__))
D = os.path.join(HERE, "golden", "jpeg_fuzz")
EXPECTED = json.load(open(os.path.join(D, "expected.json")))
NAMES = sorted(EXPECTED)


def same_float(got, want):
    if want == "nan":
        return math.isnan(got)
    if want in ("inf", "-inf"):
        return math.isinf(got) and (got > 0) == (want == "inf")
    return np.float32(got) == np.float32(want)


def check_frame(f, e, where):
    n = f.mcus_per_row * f.mcus_per_col * f.blocks_per_mcu
    assert (f.width, f.height, f.comps, f.scan_type, n) == (e["width"], e["height"], e["comps"], e["scan_type"], e["blocks"]), where
    co = np.ctypeslib.as_array(f.coeffs, (n, 64)); mz = np.ctypeslib.as_array(f.max_zag, (n,))
    assert hashlib.sha256(np.ascontiguousarray(co).tobytes()).hexdigest() == e["coeffs_sha256"], where
    assert hashlib.sha256(np.ascontiguousarray(mz).tobytes()).hexdigest() == e["max_zag_sha256"], where
    assert same_float(f.pixel_aspect_ratio, e["pixel_aspect_ratio"]) and same_float(f.dpi_y, e["dpi_y"]), (where, f.pixel_aspect_ratio, f.dpi_y)


def test_every_file_of_the_directory_has_an_expectation():
    assert sorted(n for n in os.listdir(D) if n.endswith(".jpg")) == NAMES
    kinds = [EXPECTED[n]["verdict"] for n in NAMES]
    assert kinds.count("image") >= 25 and kinds.count("null") >= 15 and kinds.count("undefined") >= 5


@pytest.mark.parametrize("name", NAMES)
def test_oracle_equals_the_second_reading(name):
    e = EXPECTED[name]
    data = open(os.path.join(D, name), "rb").read()
    buf = np.frombuffer(data, np.uint8)
    f = O.JpegFrame()
    rc = O.lib().orc_jpeg_decode_coeffs(O._ptr(buf), buf.size, C.byref(f))
    assert rc == {"image": 0, "null": -1, "undefined": -2}[e["verdict"]], (name, rc)        # -2: the reference has no defined result, the oracle refuses
    if rc == 0:
        check_frame(f, e, name)
        O.lib().orc_jpeg_frame_free(C.byref(f))
    got = O.decompress_jpeg(data, 4)                          # the whole driver (jpegload.d:3720-3808)
    assert (got is not None) == (e["verdict"] == "image")
    if got is not None:
        assert same_float(got[2], e["pixel_aspect_ratio"]) and same_float(got[3], e["dpi_y"])


@pytest.mark.parametrize("name", NAMES)
def test_host_feeder_equals_the_second_reading(name):
    e = EXPECTED[name]
    data = open(os.path.join(D, name), "rb").read()
    buf = np.frombuffer(data, np.uint8)
    L = _capi.lib()
    f = _capi.JpegFrame()
    rc = L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, buf.size, C.byref(f))
    assert (rc == _capi.OK) == (e["verdict"] == "image"), (name, rc, _capi.last_error())
    if rc == _capi.OK:
        check_frame(f, e, name)
        L.gamut_hip_jpeg_frame_free(C.byref(f))
    else:
        assert rc == _capi.ERR_DECODE and not f.coeffs and not f.max_zag and _capi.last_error()
    # the header call agrees wherever the verdict falls in front of the first scan's data
    h = _capi.JpegFrame()
    hrc = L.gamut_hip_jpeg_read_header(buf.ctypes.data, buf.size, C.byref(h))
    if e["verdict"] == "image":
        assert hrc == _capi.OK and (h.width, h.height, h.comps) == (e["width"], e["height"], e["comps"])


def test_the_probes_of_the_round_4_review():
    """VERDICT r04, "what's missing" 1: an APP1 / EXIF segment with X 300 / Y 150 dpi, unit 2 -> pixelAspectRatio 2.0, dotsPerInchY 150; byte order
    `XX` -> null.  Oracle and host feeder, II and MM, unit 3, version != 42, an IFD offset behind the segment (jpegload.d:1704-1816)."""
    L = _capi.lib()
    for name, want in (("exif_ii_300x150_r05.jpg", (2.0, 150.0)), ("exif_mm_300x150_r05.jpg", (2.0, 150.0)), ("exif_cm_r05.jpg", (np.float32(118.1 / 59.0), np.float32(np.float32(5900.0) / np.float32(39.37007874)))),
                       ("exif_bad_byte_order_r05.jpg", None), ("exif_version_43_r05.jpg", None), ("exif_ifd_offset_behind_segment_r05.jpg", None)):
        data = open(os.path.join(D, name), "rb").read()
        buf = np.frombuffer(data, np.uint8)
        got = O.decompress_jpeg(data, 3)
        f = _capi.JpegFrame()
        rc = L.gamut_hip_jpeg_decode_coeffs(buf.ctypes.data, buf.size, C.byref(f))
        if want is None:
            assert got is None and rc == _capi.ERR_DECODE, name
        else:
            assert got is not None and rc == _capi.OK, name
            assert (np.float32(got[2]), np.float32(got[3])) == (np.float32(want[0]), np.float32(want[1])), (name, got[2:], want)
            assert (np.float32(f.pixel_aspect_ratio), np.float32(f.dpi_y)) == (np.float32(want[0]), np.float32(want[1])), name
            L.gamut_hip_jpeg_frame_free(C.byref(f))
