"""GPU tests of the `Image` mirror: loadFromMemory (JPEG / PNG) and convertTo through the GPU kernels, compared with the
oracle composed the way the reference composes its pieces (plugins/jpeg.d:42-104, plugins/png.d:44-163, image.d:1180-1332)."""
import fixtures
import hashlib
import json
import os

import numpy as np
import pytest

import gen
import oracle_lib as O
from gamut_amd import image as gi
from gamut_amd.image import Image
from gamut_amd import _capi
from oracle_lib import PIXEL_TYPES, PT, PT_SIZE

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
L = gi.lib()


def expected_load(data, flags, fmt):
    """what the reference computes: codec output with the requested component count, then convertTo(applyLoadFlags)"""
    req = L.gamut_compute_requested_image_components(flags)
    if fmt == "jpeg":
        if req == 2:
            req = -1
        out, actual, par, dpi = O.decompress_jpeg(data, req)
        comps = actual if req == -1 else req
        t0 = {1: "l8", 3: "rgb8", 4: "rgba8"}[comps]
        px = out.reshape(-1)
        h = out.shape[0]; w = out.shape[1] // comps
    elif fmt == "qoi":
        r = 0 if req in (-1, 1, 2) else req                               # plugins/qoi.d:81-83
        out, fc, cs = O.qoi_decode(data, r)
        comps = fc if r == 0 else r
        t0 = {3: "rgb8", 4: "rgba8"}[comps]
        px = out.reshape(-1)
        h = out.shape[0]; w = out.shape[1] // comps
    else:
        is16 = O.png_parse(data)["depth"] == 16
        to16 = is16
        if flags & gi.LOAD_8BIT: to16 = False
        if flags & gi.LOAD_16BIT: to16 = True
        arr, n = O.stbi_load(data, 0 if req == -1 else req, to16)
        h, w, comps = arr.shape
        t0 = {1: "l", 2: "la", 3: "rgb", 4: "rgba"}[comps] + ("16" if to16 else "8")
        px = arr.reshape(-1)
    t1 = L.gamut_apply_load_flags(PT[t0], flags)
    conv = O.scanlines_convert(t0, px, t1, w, h) if t1 != PT[t0] else np.ascontiguousarray(px).view(np.uint8).reshape(-1)
    return w, h, t1, conv.reshape(h, -1)


def test_config1_640x480_jpeg_to_rgba8(hip):
    """BASELINE.json configs[0]: single 640x480 baseline 4:2:0 JPEG -> rgba8 via Image.loadFromMemory with the
    package.d:178-199 flag idiom; pixels equal the frozen golden hash"""
    data = open(os.path.join(G, "jpeg", "cfg1_640x480_420_q90.jpg"), "rb").read()
    flags = gi.LOAD_RGB | gi.LOAD_ALPHA | gi.LOAD_8BIT | gi.LOAD_NO_PREMUL | gi.LAYOUT_VERT_STRAIGHT | gi.LAYOUT_GAPLESS
    im = Image()
    assert im.loadFromMemory(data, flags), im.errorMessage
    assert (im.width, im.height, PIXEL_TYPES[im.type], im.layers) == (640, 480, "rgba8", 1)
    assert im.pitchInBytes == 640 * 4 and im.isOwned and not im.isStoredUpsideDown
    golden = json.load(open(os.path.join(G, "golden.json")))["frozen"]["cfg1_640x480_420_q90:comps4"]
    assert hashlib.sha256(im.pixels().tobytes()).hexdigest() == golden


JPEGS = fixtures.jpegs("s_*.jpg", at_least=6)
FLAGSETS = [0, gi.LOAD_RGB | gi.LOAD_ALPHA | gi.LOAD_8BIT, gi.LOAD_GREYSCALE | gi.LOAD_NO_ALPHA, gi.LOAD_GREYSCALE | gi.LOAD_ALPHA,
            gi.LOAD_FP32 | gi.LOAD_GREYSCALE, gi.LOAD_16BIT | gi.LOAD_RGB | gi.LOAD_NO_ALPHA, gi.LOAD_RGB | gi.LOAD_ALPHA | gi.LOAD_PREMUL | gi.LOAD_FP32]


@pytest.mark.parametrize("path", JPEGS, ids=[os.path.basename(p) for p in JPEGS])
def test_load_jpeg_with_flags(hip, path):
    data = open(path, "rb").read()
    for flags in FLAGSETS:
        for layout in (0, gi.LAYOUT_VERT_FLIPPED | gi.LAYOUT_ALIGNED[64], gi.LAYOUT_GAPLESS):
            w, h, t1, exp = expected_load(data, flags, "jpeg")
            im = Image()
            assert im.loadFromMemory(data, flags | layout), im.errorMessage
            assert (im.width, im.height, im.type) == (w, h, t1)
            assert np.array_equal(im.pixels(), exp), f"flags={flags:#x} layout={layout}"
            if layout & gi.LAYOUT_VERT_FLIPPED:
                assert im.isStoredUpsideDown and im.scanptr(0) % 64 == 0 and im.pitchInBytes % 64 == 0
    im = Image()
    assert not im.loadFromMemory(data, gi.LOAD_RGB | gi.LOAD_GREYSCALE) and im.errorMessage == "Invalid image decoding flags"
    assert not im.loadFromMemory(data[:300], 0) and im.errorMessage == "Image decoding failed"
    if path.endswith("issue35.jpg"):
        assert im.loadFromMemory(data) and im.pixelAspectRatio == 1.0 and im.dotsPerInchY == 72.0


def test_load_jpeg_density_and_verdicts_of_the_input_layer(hip):
    """Image.loadFromMemory on the files of tests/golden/jpeg_fuzz (plugins/jpeg.d:61-100): an EXIF segment's resolution reaches pixelAspectRatio /
    dotsPerInchY (2.0 / 150 for the round-4 review's probe), a malformed one fails the load, a file without any density hands on the NaN the D
    struct's members start with (`== -1` at :96-97 does not catch it), a file that ends inside its scan loads."""
    import json
    import math
    d = os.path.join(G, "jpeg_fuzz")
    expected = json.load(open(os.path.join(d, "expected.json")))
    for name, e in sorted(expected.items()):
        data = open(os.path.join(d, name), "rb").read()
        if data[:2] != b"\xff\xd8":
            continue                                          # detectJPEG (plugins/jpeg.d:103-107) wants the signature at offset 0: not a JPEG for loadFromMemory
        im = Image()
        ok = im.loadFromMemory(data, gi.LOAD_RGB | gi.LOAD_8BIT)
        assert ok == (e["verdict"] == "image"), (name, im.errorMessage)
        if not ok:
            assert im.errorMessage == "Image decoding failed", name
            continue
        assert (im.width, im.height) == (e["width"], e["height"])
        for got, want in ((im.pixelAspectRatio, e["pixel_aspect_ratio"]), (im.dotsPerInchY, e["dpi_y"])):
            assert math.isnan(got) if want == "nan" else np.float32(got) == np.float32(want), (name, got, want)
        want_px = O.decompress_jpeg(data, 3)[0]
        assert np.array_equal(im.pixels()[:, :e["width"] * 3], want_px), name
    im = Image()
    assert im.loadFromMemory(open(os.path.join(d, "exif_ii_300x150_r05.jpg"), "rb").read()) and (im.pixelAspectRatio, im.dotsPerInchY) == (2.0, 150.0)
    assert not im.loadFromMemory(open(os.path.join(d, "exif_bad_byte_order_r05.jpg"), "rb").read())


def test_load_png_with_flags(hip):
    rng = np.random.default_rng(4)
    w, h = 21, 13
    files = {"issue76": open(os.path.join(G, "ref_images", "issue76.png"), "rb").read(),
             "vst3": open(os.path.join(G, "ref_images", "vst3-compatible.png"), "rb").read(),
             "rgb8": gen.write_png(rng.integers(0, 256, (h, w * 3)), w, h, 2, 8),
             "rgba16": gen.write_png(rng.integers(0, 65536, (h, w * 4)), w, h, 6, 16),
             "pal4": gen.write_png(rng.integers(0, 16, (h, w)), w, h, 3, 4, palette=rng.integers(0, 256, (16, 3)), trns=[0, 128, 255]),
             "la8": gen.write_png(rng.integers(0, 256, (h, w * 2)), w, h, 4, 8, interlace=1)}
    for name, data in files.items():
        for flags in FLAGSETS + [gi.LOAD_8BIT, gi.LOAD_16BIT]:
            ew, eh, t1, exp = expected_load(data, flags, "png")
            im = Image()
            assert im.loadFromMemory(data, flags | gi.LAYOUT_TRAILING[3]), f"{name}: {im.errorMessage}"
            assert (im.width, im.height, im.type) == (ew, eh, t1), name
            assert np.array_equal(im.pixels(), exp), f"{name} flags={flags:#x}"
    im = Image()                                              # testIssue76 (test-suite main.d:172-190)
    assert im.loadFromMemory(files["issue76"]) and PIXEL_TYPES[im.type] == "l16" and (im.width, im.height) == (2, 2)
    assert im.scanline(0).view(np.uint16).tolist() == [1875, 65535] and im.scanline(1).view(np.uint16).tolist() == [0, 2807]


def test_issue65_sequence(hip):
    """examples/test-suite/source/main.d testIssue65: load FP32|GREYSCALE, setLayout(TRAILING_1), setLayout(TRAILING_0), convertTo8Bit"""
    data = open(os.path.join(G, "ref_images", "issue65.png"), "rb").read()
    im = Image()
    assert im.loadFromMemory(data, gi.LOAD_FP32 | gi.LOAD_GREYSCALE) and im.hasData and im.isValid and PIXEL_TYPES[im.type] == "laf32"
    assert im.setLayout(gi.LAYOUT_TRAILING[1]) and abs(im.pitchInBytes) >= (1024 + 1) * 8
    assert im.setLayout(gi.LAYOUT_TRAILING[0]) and im.hasData
    assert im.convertTo8Bit() and im.hasData and PIXEL_TYPES[im.type] == "la8"
    arr, n = O.stbi_load(data, 0)
    f = O.scanlines_convert("rgba8", arr.reshape(-1), "laf32", 1024, 1024)
    exp = O.scanlines_convert("laf32", f, "la8", 1024, 1024)
    assert np.array_equal(im.pixels().reshape(-1), exp)


def test_convert_to_layouts_and_layers(hip):
    """convertTo on layered images with borders / alignment / flips: every layer converted, gap bytes irrelevant"""
    rng = np.random.default_rng(9)
    w, h, layers = 11, 6, 3
    for src, dst, lay in [("rgba8", "rgbaf32", gi.LAYOUT_BORDER[1] | gi.LAYOUT_ALIGNED[32]), ("rgbaf32", "rgb16", gi.LAYOUT_VERT_FLIPPED),
                          ("rgb16", "l8", gi.LAYOUT_GAPLESS), ("l8", "la16", gi.LAYOUT_TRAILING[7] | gi.LAYOUT_MULTIPLICITY[8]),
                          ("la16", "la16", gi.LAYOUT_VERT_FLIPPED | gi.LAYOUT_ALIGNED[128])]:
        im = Image()
        assert im.createLayered(w, h, layers, PT[src])
        px = gen.make_pixels(src, w * h * layers, rng).reshape(layers, h, -1).view(np.uint8).reshape(layers, h, -1)
        for l in range(layers):
            for y in range(h):
                row = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint8 * px.shape[2]).from_address(im.layerptr(l, y)))
                row[:] = px[l, y]
        assert im.convertTo(PT[dst], lay), im.errorMessage
        assert im.type == PT[dst] and im.layers == layers and im.layoutConstraints == lay
        for l in range(layers):
            exp = O.scanlines_convert(src, px[l].reshape(-1), dst, w, h).reshape(h, -1) if src != dst else px[l]
            assert np.array_equal(im.pixels(l), exp), f"{src}->{dst} layer {l}"


def test_load_qoi_with_flags_and_mixed_batch(hip):
    """QOI through Image.loadFromMemory with every flag set (plugins/qoi.d:47-141), then BASELINE.json config 5 in miniature:
    a mixed JPEG / PNG / QOI list, image i owned by rank i % N (gamut_amd.shard), every owner's result == the oracle's."""
    from gamut_amd.shard import shard_indices, owner_of
    from test_oracle_pinning import _qoi_test_images
    qois = [gen.qoi_encode(a) for a in _qoi_test_images()[1:4]]
    for data in qois:
        for flags in FLAGSETS:
            w, h, t1, exp = expected_load(data, flags, "qoi")
            im = Image()
            assert im.loadFromMemory(data, flags | gi.LAYOUT_TRAILING[1]), im.errorMessage
            assert (im.width, im.height, im.type) == (w, h, t1)
            assert np.array_equal(im.pixels(), exp), f"flags={flags:#x}"
    assert gi.lib().gamut_identify_format_from_memory(qois[0], len(qois[0])) == 2          # ImageFormat.QOI
    rng = np.random.default_rng(9)
    batch = []
    for i in range(12):
        kind = ("jpeg", "png", "qoi")[i % 3]
        if kind == "jpeg":
            data = open(JPEGS[i % len(JPEGS)], "rb").read()
        elif kind == "png":
            data = gen.write_png(rng.integers(0, 256, (9 + i, (20 + i) * 4)), 20 + i, 9 + i, 6, 8)
        else:
            data = qois[i % len(qois)]
        batch.append((kind, data))
    for world in (1, 2, 8):
        seen = []
        for rank in range(world):
            for i in shard_indices(len(batch), rank, world):
                assert owner_of(i, world) == rank
                kind, data = batch[i]
                w, h, t1, exp = expected_load(data, gi.LOAD_RGB | gi.LOAD_ALPHA | gi.LOAD_8BIT, kind)
                im = Image()
                assert im.loadFromMemory(data, gi.LOAD_RGB | gi.LOAD_ALPHA | gi.LOAD_8BIT), im.errorMessage
                assert PIXEL_TYPES[im.type] == "rgba8" and np.array_equal(im.pixels(), exp), (kind, i)
                seen.append(i)
        assert sorted(seen) == list(range(len(batch)))


def test_device_resident_images(hip):
    """the storage extension (gamut_image_set_device_storage): the same loads, flags, layouts and convertTo chains with the
    pixels in HBM give the same bytes as the host-storage mirror and the oracle"""
    from test_oracle_pinning import _qoi_test_images
    rng = np.random.default_rng(17)
    files = [("jpeg", open(p, "rb").read()) for p in JPEGS[:3]]
    files.append(("jpeg", open(os.path.join(G, "jpeg", "p_131x97_420.jpg"), "rb").read()))                     # progressive: host feeder + upload
    files.append(("png", open(os.path.join(G, "ref_images", "issue76.png"), "rb").read()))
    files.append(("png", gen.write_png(rng.integers(0, 16, (13, 21)), 21, 13, 3, 4, palette=rng.integers(0, 256, (16, 3)), trns=[0, 128, 255])))
    files.append(("png", gen.write_png(rng.integers(0, 65536, (13, 21 * 4)), 21, 13, 6, 16)))
    files.append(("qoi", gen.qoi_encode(_qoi_test_images()[3])))
    for kind, data in files:
        for flags in FLAGSETS[:5]:
            for layout in (0, gi.LAYOUT_VERT_FLIPPED | gi.LAYOUT_ALIGNED[64], gi.LAYOUT_TRAILING[3] | gi.LAYOUT_BORDER[1]):
                w, h, t1, exp = expected_load(data, flags, kind)
                im = Image(device=True)
                assert im.loadFromMemory(data, flags | layout), (kind, im.errorMessage)
                assert im.isDevice and (im.width, im.height, im.type) == (w, h, t1)
                assert np.array_equal(im.pixels(), exp), (kind, hex(flags), layout)
                host = Image(); assert host.loadFromMemory(data, flags | layout)
                assert (im.pitchInBytes, im.layoutConstraints, im.isStoredUpsideDown) == (host.pitchInBytes, host.layoutConstraints, host.isStoredUpsideDown)
    # create + convertTo chain + flip, layered
    a = Image(device=True); b = Image()
    for im in (a, b):
        assert im.createLayered(33, 9, 3, PT["rgba8"], gi.LAYOUT_TRAILING[1])
        assert im.convertTo(PT["rgbaf32"], gi.LAYOUT_GAPLESS) and im.convertTo(PT["la16"], 0) and im.flipVertical()
    for layer in range(3):
        assert np.array_equal(a.pixels(layer), b.pixels(layer))
    assert not Image(device=True).loadFromMemory(b"garbage", 0)


@pytest.mark.parametrize("device", [False, True], ids=["host", "hbm"])
def test_flips_clone_and_copy(hip, device):
    """flipHorizontal (image.d:1475-1509), flipVertical under a vertical constraint = flipVerticalPhysical (:1926-1954), clone /
    copyPixelsTo (:795-841), on layered images of every pixel size, host and HBM storage, straight and upside-down storage"""
    rng = np.random.default_rng(4)
    for tname, w, h, layers, layout in [("l8", 7, 5, 1, 0), ("rgb8", 6, 4, 3, gi.LAYOUT_ALIGNED[16] | gi.LAYOUT_BORDER[1]), ("rgba8", 33, 9, 2, gi.LAYOUT_VERT_FLIPPED),
                                        ("la16", 5, 2, 2, gi.LAYOUT_VERT_STRAIGHT), ("rgb16", 4, 3, 1, gi.LAYOUT_TRAILING[3]), ("rgbaf32", 9, 7, 2, gi.LAYOUT_GAPLESS),
                                        ("rgbf32", 3, 1, 1, 0), ("rgba16", 1, 6, 2, 0)]:
        t = PT[tname]
        src = [rng.integers(0, 256, (h, w * O.PT_SIZE[t]), dtype=np.uint8) for _ in range(layers)]
        host = Image()
        views = []
        for a in src:                                            # fill through convertTo-free means: a view per layer copied in
            v = Image(); assert v.createView(a, w, h, t, a.shape[1]); views.append(v)
        im = Image(device=device)
        assert im.createLayered(w, h, layers, t, layout)
        for k, v in enumerate(views):
            stage = Image(device=device)
            assert stage.createLayered(w, h, 1, t, 0)
            if device:
                import ctypes as C
                for y in range(h):
                    _capi.check(hip.gamut_hip_memcpy_h2d(stage.scanptr(y), src[k][y].ctypes.data, src[k].shape[1], None))
                _capi.check(hip.gamut_hip_stream_synchronize(None))
            else:
                assert v.copyPixelsTo(stage)
            assert stage.copyPixelsTo(im.layer(k))
        px = lambda img: [img.pixels(k) for k in range(layers)]
        assert all(np.array_equal(a, b) for a, b in zip(px(im), src))
        c = im.clone()
        assert c.isValid and c.isOwned and c.layoutConstraints == im.layoutConstraints and c.scanptr(0) != im.scanptr(0) and c.isDevice == device
        assert all(np.array_equal(a, b) for a, b in zip(px(c), src))
        ps = O.PT_SIZE[t]
        assert im.flipHorizontal()
        assert all(np.array_equal(a, b.reshape(h, w, ps)[:, ::-1].reshape(h, w * ps)) for a, b in zip(px(im), src))
        assert all(np.array_equal(a, b) for a, b in zip(px(c), src)), "the clone shares nothing with its source"
        p0, pitch0 = im.scanptr(0), im.pitchInBytes
        assert im.flipVertical()
        exp = [b.reshape(h, w, ps)[::-1, ::-1].reshape(h, w * ps) for b in src]
        assert all(np.array_equal(a, b) for a, b in zip(px(im), exp))
        if layout & (gi.LAYOUT_VERT_FLIPPED | gi.LAYOUT_VERT_STRAIGHT):
            assert im.scanptr(0) == p0 and im.pitchInBytes == pitch0, "a vertical constraint pins the storage order: the rows moved"
        else:
            assert im.pitchInBytes == -pitch0
        other = Image(device=device)
        assert other.createLayered(w + 1, h, layers, t, 0) and not im.copyPixelsTo(other)           # size mismatch (an assert in the reference)


def test_concurrent_host_threads(hip):
    """the library keeps no mutable global state besides per-thread streams and staging buffers: decodes of
    different Images from several host threads at once give the single-threaded results (ctypes drops the GIL in the calls)"""
    import threading
    from test_oracle_pinning import _qoi_test_images
    rng = np.random.default_rng(23)
    work = [("jpeg", open(p, "rb").read(), 0) for p in JPEGS[:4]]
    work += [("jpeg", open(os.path.join(G, "jpeg", "p_131x97_420.jpg"), "rb").read(), gi.LOAD_RGB | gi.LOAD_ALPHA)]
    work += [("png", open(os.path.join(G, "ref_images", "issue65.png"), "rb").read(), gi.LOAD_FP32),
             ("png", gen.write_png(rng.integers(0, 256, (40, 57 * 3)), 57, 40, 2, 8), gi.LOAD_RGB | gi.LOAD_ALPHA | gi.LOAD_16BIT),
             ("qoi", gen.qoi_encode(_qoi_test_images()[3]), 0)]
    expected = [expected_load(d, f, k) for k, d, f in work]
    errors = []

    def worker(tid):
        try:
            for rep in range(6):
                for j in range(len(work)):
                    k, d, f = work[(j + tid) % len(work)]
                    w, h, t1, exp = expected[(j + tid) % len(work)]
                    im = Image(device=(tid + rep) % 2 == 1)
                    if not im.loadFromMemory(d, f):
                        errors.append((tid, k, im.errorMessage)); return
                    if (im.width, im.height, im.type) != (w, h, t1) or not np.array_equal(im.pixels(), exp):
                        errors.append((tid, k, "pixels differ")); return
        except Exception as e:            # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors[:3]
