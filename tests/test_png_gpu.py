"""GPU parity: PNG de-filter / expand and the whole stb load path through the C ABI vs the CPU oracle
(and vs the pixels the streams were built from: PNG is lossless).  Bar: bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import gen
import oracle_lib as O
from gamut_amd import _capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def up(L, arr):
    arr = np.ascontiguousarray(arr)
    p = L.gamut_hip_device_malloc(max(16, arr.nbytes))
    assert p
    _capi.check(L.gamut_hip_memcpy_h2d(p, arr.ctypes.data, arr.nbytes, None))
    _capi.check(L.gamut_hip_stream_synchronize(None))
    return p


def down(L, p, n):
    out = np.empty(n, np.uint8)
    _capi.check(L.gamut_hip_stream_synchronize(None))
    _capi.check(L.gamut_hip_memcpy_d2h(out.ctypes.data, p, n, None))
    _capi.check(L.gamut_hip_stream_synchronize(None))
    return out


def gpu_defilter(L, raw, x, y, img_n, out_n, depth, color, count=1, raw_stride=0, expect_status=0):
    raw = np.ascontiguousarray(raw, np.uint8)
    nbytes = x * y * out_n * (2 if depth == 16 else 1)
    ostride = nbytes + 64
    draw = up(L, raw)
    dout = up(L, np.full(count * ostride, 0xA5, np.uint8))
    dst = up(L, np.zeros(count, np.uint32))
    raw_len = raw.size if count == 1 else raw_stride
    _capi.check(L.gamut_hip_png_defilter_batch_device(draw, raw_stride, raw_len, dout, ostride, x, y, img_n, out_n, depth, color,
                                                       count, dst, None))
    host = down(L, dout, count * ostride)
    status = down(L, dst, count * 4).view(np.uint32)
    for p in (draw, dout, dst):
        L.gamut_hip_device_free(p)
    outs = []
    for i in range(count):
        img = host[i * ostride:(i + 1) * ostride]
        assert (img[nbytes:] == 0xA5).all(), "wrote past the image"
        outs.append(img[:nbytes].copy())
    assert (status != 0).astype(int).tolist() == ([expect_status] * count if np.isscalar(expect_status) else list(expect_status))
    return outs


@pytest.fixture(params=["workgroups", "queue", "workgroups+aligned", "queue+aligned", "queue+roll"])
def launch_mode(request):
    """the two launch shapes of the ring kernels (png.hip): one workgroup per image / row segment, or every (image, band) unit
    of the batch through the device-wide work queue (the launcher's own rule picks the queue for large, badly dividing batches);
    each with the row-aligned loads of round 1-3 and with the line-aligned loads of round 4 (forced on for every row of at least one
    piece: the launcher's own rule takes them from 256-byte rows on); and the queue in its rolling form (round 5: rings of eight pieces,
    a transfer per trip -- forced on for every row of at least one piece too)"""
    shape, _, al = request.param.partition("+")
    old = {k: os.environ.get(k) for k in ("GAMUT_HIP_PNG_QUEUE", "GAMUT_HIP_PNG_ALIGNED", "GAMUT_HIP_PNG_ROLL")}
    os.environ["GAMUT_HIP_PNG_QUEUE"] = "1" if shape == "queue" else "0"
    os.environ["GAMUT_HIP_PNG_ALIGNED"] = "1" if al == "aligned" else "0"
    os.environ["GAMUT_HIP_PNG_ROLL"] = "1" if al == "roll" else "0"
    yield request.param
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


FORMATS = [(1, 1, 0), (1, 2, 0), (1, 4, 0), (1, 8, 0), (1, 16, 0), (2, 8, 4), (2, 16, 4), (3, 8, 2), (3, 16, 2), (4, 8, 6), (4, 16, 6),
           (1, 1, 3), (1, 2, 3), (1, 4, 3), (1, 8, 3)]


@pytest.mark.parametrize("img_n,depth,color", FORMATS)
def test_defilter_formats_filters_sizes(hip, launch_mode, img_n, depth, color):
    rng = np.random.default_rng(depth * 10 + img_n)
    fb = 1 if depth < 8 else img_n * (2 if depth == 16 else 1)
    for (x, y) in [(1, 1), (2, 3), (5, 2), (7, 5), (37, 70), (64, 65), (130, 129), (259, 67), (33, 1100)]:
        smooth = rng.integers(0, 1 << depth, (y, x * img_n))
        smooth = (np.cumsum(rng.integers(-2, 3, (y, x * img_n)), axis=1) + smooth[:, :1]) % (1 << depth) if x > 8 else smooth
        rows = gen.pack_samples(smooth, depth)
        fsets = [np.full(y, t, np.uint8) for t in range(5)] + [rng.integers(0, 5, y).astype(np.uint8), gen.png_heuristic_filters(rows, fb)]
        if (x, y) in ((33, 1100),):
            fsets = fsets[4:]
        for filt in fsets:
            raw = gen.png_forward_filter(rows, fb, filt)
            for out_n in ([img_n] + ([img_n + 1] if img_n in (1, 3) and color != 3 else [])):
                exp = O.png_create_image_raw(raw, img_n, out_n, x, y, depth, color)
                assert exp is not None
                got = gpu_defilter(hip, raw, x, y, img_n, out_n, depth, color)[0]
                assert np.array_equal(got, exp), f"n={img_n} d={depth} {x}x{y} out_n={out_n} filters={np.unique(filt)}: {np.count_nonzero(got != exp)} bytes differ"


@pytest.mark.parametrize("x,y", [(1004, 131), (962, 200), (1001, 70), (3848, 130)])
def test_rows_off_the_memory_lines(hip, launch_mode, x, y):
    """tight RGBA8 / RGB8 rows of widths that are no multiple of 32 pixels: the rows start anywhere in a 128-byte line (pitches 4016 / 3848 / 4004 /
    15392 bytes) and the write-back groups are the lines of memory, not the row's own pieces (png.hip: wb_lines) -- several bands (the hand-off
    publishes what the LINE groups have stored), several images at a stride that moves every image's first line, every filter mix."""
    rng = np.random.default_rng(x + y)
    for img_n, out_n, color in ((4, 4, 6), (3, 4, 2), (3, 3, 2)):
        n = 3
        px = [(np.cumsum(rng.integers(-3, 4, (y, x * img_n)), axis=1) + rng.integers(0, 256, (y, 1))) % 256 for _ in range(n)]
        for filt in (rng.integers(0, 5, y).astype(np.uint8), np.full(y, 4, np.uint8), gen.png_heuristic_filters(px[0], img_n)):
            stride = (x * img_n + 1) * y + 3
            raws = np.zeros(n * stride, np.uint8); exps = []
            for i in range(n):
                r = gen.png_forward_filter(px[i], img_n, filt)
                raws[i * stride:i * stride + r.size] = r
                exps.append(O.png_create_image_raw(r, img_n, out_n, x, y, 8, color))
            got = gpu_defilter(hip, raws, x, y, img_n, out_n, 8, color, count=n, raw_stride=stride)
            for i in range(n):
                assert np.array_equal(got[i], exps[i]), (x, y, img_n, out_n, i, np.unique(filt), int(np.count_nonzero(got[i] != exps[i])))


def test_config3_geometry_roundtrip(hip):
    """3840x2160 RGBA8 (BASELINE.json config 3): forward-filter -> GPU de-filter gives the pixels back; both filter policies."""
    w, h = 3840, 2160
    img = gen.synth_rgb(w, h, 77, alpha=True).reshape(h, w * 4)
    for name, filt in [("heuristic", gen.png_heuristic_filters(img, 4)), ("random", np.random.default_rng(3).integers(0, 5, h).astype(np.uint8)),
                       ("paeth", np.full(h, 4, np.uint8))]:
        raw = gen.png_forward_filter(img, 4, filt)
        got = gpu_defilter(hip, raw, w, h, 4, 4, 8, 6)[0]
        assert np.array_equal(got, img.reshape(-1)), name
    exp = O.png_create_image_raw(raw, 4, 4, w, h, 8, 6)
    assert np.array_equal(exp, img.reshape(-1))


@pytest.mark.parametrize("img_n,out_n,color", [(4, 4, 6), (3, 4, 2), (3, 3, 2), (1, 1, 0)])
def test_row_segments_of_small_batches(hip, img_n, out_n, color):
    """small batches cut every image into row segments at None / Sub rows (png.hip: k_png_segments): rows without any cut row,
    a single cut row, cut rows everywhere, cut rows only far from the targets; 1 .. 3 images a call"""
    rng = np.random.default_rng(40 + img_n)
    for (x, y) in [(97, 256), (64, 1000), (301, 2050)]:
        px = rng.integers(0, 256, (y, x * img_n))
        patterns = [np.full(y, 4, np.uint8), np.full(y, 2, np.uint8), rng.integers(0, 5, y).astype(np.uint8), rng.integers(2, 5, y).astype(np.uint8)]
        one = np.full(y, 3, np.uint8); one[y // 2 + 7] = 1; patterns.append(one)
        far = rng.integers(2, 5, y).astype(np.uint8); far[3] = 0; far[y - 2] = 1; far[y // 8 + y // 20] = 1; patterns.append(far)
        for filt in patterns:
            raw = gen.png_forward_filter(px, img_n, filt)
            exp = O.png_create_image_raw(raw, img_n, out_n, x, y, 8, color)
            assert np.array_equal(gpu_defilter(hip, raw, x, y, img_n, out_n, 8, color)[0], exp), (x, y, np.unique(filt))
        n = 3
        stride = (x * img_n + 1) * y + 5
        raws = np.zeros(n * stride, np.uint8); exps = []
        for i in range(n):
            r = gen.png_forward_filter(rng.integers(0, 256, (y, x * img_n)), img_n, rng.integers(0, 5, y).astype(np.uint8))
            raws[i * stride:i * stride + r.size] = r
            exps.append(O.png_create_image_raw(r, img_n, out_n, x, y, 8, color))
        got = gpu_defilter(hip, raws, x, y, img_n, out_n, 8, color, count=n, raw_stride=stride)
        for i in range(n):
            assert np.array_equal(got[i], exps[i]), (x, y, i)


@pytest.mark.parametrize("img_n,out_n,depth,color,x,y", [(4, 4, 8, 6, 333, 647), (3, 4, 8, 2, 333, 647), (3, 3, 8, 2, 171, 700), (2, 2, 16, 4, 150, 333),
                                                         (4, 4, 16, 6, 67, 1100), (1, 1, 8, 0, 2048, 200)])
def test_work_queue_batches(hip, img_n, out_n, depth, color, x, y):
    """k_png_defilter_queue: 300 images (more (image, band) units than the chip has wave slots) whose bands run at very different
    speeds -- rows all None next to rows all Paeth, so consumers catch up with their producers and wait on them -- ragged widths,
    a partial last band, chains of Up / Avg / Paeth rows that never cut; the hand-off of a band's last row goes from whichever
    compute unit drew the band above to whichever drew this one.  Every image is checked byte for byte, three launches in a row
    (the queue state is re-zeroed by every launch), then once more under the launcher's own rule."""
    rng = np.random.default_rng(img_n * 100 + depth)
    fb = img_n * (2 if depth == 16 else 1)
    kinds = [np.full(y, 4, np.uint8), np.full(y, 0, np.uint8), np.full(y, 2, np.uint8), np.full(y, 3, np.uint8), rng.integers(0, 5, y).astype(np.uint8),
             rng.integers(2, 5, y).astype(np.uint8), np.where(np.arange(y) % 64 == 63, 4, 1).astype(np.uint8)]
    raws, exps = [], []
    for filt in kinds:
        rows = gen.pack_samples(rng.integers(0, 1 << depth, (y, x * img_n)), depth)
        raw = gen.png_forward_filter(rows, fb, filt)
        raws.append(raw); exps.append(O.png_create_image_raw(raw, img_n, out_n, x, y, depth, color))
    n = 300
    stride = raws[0].size + 3
    order = rng.integers(0, len(kinds), n)
    order[:len(kinds)] = np.arange(len(kinds))
    batch = np.zeros(n * stride, np.uint8)
    for i, k in enumerate(order):
        batch[i * stride:i * stride + raws[k].size] = raws[k]
    old = os.environ.get("GAMUT_HIP_PNG_QUEUE")
    try:
        for mode in ("1", "1", "1", None):
            if mode is None:
                os.environ.pop("GAMUT_HIP_PNG_QUEUE", None)
            else:
                os.environ["GAMUT_HIP_PNG_QUEUE"] = mode
            got = gpu_defilter(hip, batch, x, y, img_n, out_n, depth, color, count=n, raw_stride=stride)
            bad = [i for i in range(n) if not np.array_equal(got[i], exps[order[i]])]
            assert not bad, f"mode {mode}: images {bad[:10]} differ (filters of the first: {np.unique(kinds[order[bad[0]]])})"
    finally:
        if old is None:
            os.environ.pop("GAMUT_HIP_PNG_QUEUE", None)
        else:
            os.environ["GAMUT_HIP_PNG_QUEUE"] = old


def test_batch_and_corrupt_filter(hip, launch_mode):
    rng = np.random.default_rng(12)
    x, y, n = 61, 150, 4
    stride = (x * 4 + 1) * y + 13
    raws = np.zeros(n * stride, np.uint8)
    exps = []
    for i in range(n):
        img = rng.integers(0, 256, (y, x * 4), dtype=np.uint8)
        raw = gen.png_forward_filter(img, 4, rng.integers(0, 5, y))
        if i == 2:
            raw[(x * 4 + 1) * 70] = 9          # invalid filter type: "Corrupt PNG" (stbdec.d:1438)
        raws[i * stride:i * stride + raw.size] = raw
        exps.append(img.reshape(-1))
    got = gpu_defilter(hip, raws, x, y, 4, 4, 8, 6, count=n, raw_stride=stride, expect_status=[0, 0, 1, 0])
    for i in (0, 1, 3):
        assert np.array_equal(got[i], exps[i])
    assert O.png_create_image_raw(raws[2 * stride:3 * stride], 4, 4, x, y, 8, 6) is None
    # not enough pixels (stbdec.d:1430)
    d = up(hip, raws)
    assert hip.gamut_hip_png_defilter_batch_device(d, 0, 100, d, 0, x, y, 4, 4, 8, 6, 1, None, None) == _capi.ERR_DECODE
    hip.gamut_hip_device_free(d)


def _load(hip, data, req, sixteen):
    buf = np.frombuffer(data, np.uint8)
    x, y, n = C.c_int(), C.c_int(), C.c_int()
    fx, fy, fr = C.c_float(), C.c_float(), C.c_float()
    fn = hip.gamut_hip_stbi_load_16_from_memory if sixteen else hip.gamut_hip_stbi_load_from_memory
    p = fn(buf.ctypes.data, buf.size, C.byref(x), C.byref(y), C.byref(n), req, C.byref(fx), C.byref(fy), C.byref(fr))
    if not p:
        return None
    comps = n.value if req == 0 else req
    ct = C.c_uint16 if sixteen else C.c_uint8
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (y.value, x.value, comps)).copy()
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]; libc.free(p)
    return out, n.value, (fx.value, fy.value, fr.value)


def _check_file(hip, data, label):
    for sixteen in (False, True):
        for req in (0, 1, 2, 3, 4):
            exp = O.stbi_load(data, req, sixteen)
            got = _load(hip, data, req, sixteen)
            if exp is None:
                assert got is None, label
                continue
            assert got is not None, f"{label} req={req} 16={sixteen}: {hip.gamut_hip_last_error()}"
            assert got[1] == exp[1] and np.array_equal(got[0], exp[0]), f"{label} req={req} 16={sixteen}"


@pytest.mark.parametrize("name", ["issue65.png", "vst3-compatible.png", "issue76.png", "issue51cgbi.png", "issue51cgbi2.png"])
def test_reference_fixture_files(hip, name):
    _check_file(hip, open(os.path.join(HERE, "golden", "ref_images", name), "rb").read(), name)


def test_truncated_reference_files_load(hip):
    """Gamut issue #92 (missing IEND / truncated CRC): must still decode (stbdec.d:2008-2012)."""
    for name in ["issue92-no-IEND.png", "issue92-truncated-in-CRC.png"]:
        data = open(os.path.join(HERE, "golden", "ref_images", name), "rb").read()
        exp = O.stbi_load(data, 4, False)
        got = _load(hip, data, 4, False)
        assert got is not None and np.array_equal(got[0], exp[0]), name


def test_generated_png_files(hip):
    """every colour type x depth, tRNS, palettes, Adam7, CgBI, pHYs, split IDATs"""
    rng = np.random.default_rng(21)
    w, h = 23, 19
    cases = []
    for color, depths in [(0, (1, 2, 4, 8, 16)), (2, (8, 16)), (4, (8, 16)), (6, (8, 16)), (3, (1, 2, 4, 8))]:
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
        for depth in depths:
            smp = rng.integers(0, 1 << depth, (h, w * ch))
            pal = rng.integers(0, 256, (1 << min(depth, 8), 3)) if color == 3 else None
            for interlace in (0, 1):
                cases.append((f"c{color}d{depth}i{interlace}", gen.write_png(smp, w, h, color, depth, interlace=interlace, palette=pal)))
            if color == 0:
                key = int(smp[0, 0])
                cases.append((f"c0d{depth}trns", gen.write_png(smp, w, h, 0, depth, trns=[key >> 8, key & 255])))
            if color == 2:
                k = [int(v) for v in smp[0, :3]]
                cases.append((f"c2d{depth}trns", gen.write_png(smp, w, h, 2, depth, trns=sum(([v >> 8, v & 255] for v in k), []))))
            if color == 3:
                cases.append((f"c3d{depth}trns", gen.write_png(smp, w, h, 3, depth, palette=pal, trns=list(rng.integers(0, 256, min(len(pal), 5))))))
    smp = rng.integers(0, 256, (h, w * 4))
    cases.append(("cgbi", gen.write_png(smp, w, h, 6, 8, iphone=True)))
    cases.append(("phys", gen.write_png(smp, w, h, 6, 8, extra_chunks=[(b"pHYs", (3780).to_bytes(4, "big") + (7560).to_bytes(4, "big") + b"\x01")])))
    cases.append(("noiend", gen.write_png(smp, w, h, 6, 8, no_iend=True)))
    # a PLTE shorter than the index range (legal): indices >= pal_len expand to (0,0,0,0), the zero-initialised tail of the
    # reference's palette array (stbdec.d:1779) -- with and without tRNS, 4- and 8-bit indices
    short_pal = rng.integers(1, 256, (5, 3))
    cases.append(("c3d4_short_plte", gen.write_png(rng.integers(0, 16, (h, w)), w, h, 3, 4, palette=short_pal)))
    cases.append(("c3d8_short_plte_trns", gen.write_png(rng.integers(0, 256, (h, w)), w, h, 3, 8, palette=short_pal, trns=[7, 200])))
    cases.append(("1x1", gen.write_png(smp[:1, :4], 1, 1, 6, 8)))
    cases.append(("adam7_small", gen.write_png(smp[:3, :8], 2, 3, 6, 8, interlace=1)))
    for label, data in cases:
        _check_file(hip, data, label)
    got = _load(hip, dict(cases)["phys"], 0, False)
    exp_info = O.png_parse(dict(cases)["phys"])
    assert got[2] == (exp_info["ppmX"], exp_info["ppmY"], exp_info["pixelAspectRatio"]) == (3780.0, 7560.0, 0.5)
    # garbage in: NULL + message out
    assert _load(hip, b"not a png at all", 0, False) is None and hip.gamut_hip_last_error() != b""
    assert _load(hip, dict(cases)["1x1"][:40], 0, False) is None


@pytest.mark.parametrize("img_n,out_n,x,y", [(3, 4, 3, 70001), (1, 1, 70003, 2), (4, 4, 5, 66000), (2, 2, 33, 65537)])
def test_extreme_geometry(hip, launch_mode, img_n, out_n, x, y):
    """more rows than a grid dimension holds (65535), rows of one piece or less, very wide two-row images"""
    rng = np.random.default_rng(x + y)
    px = rng.integers(0, 256, (y, x * img_n)).astype(np.uint8)
    raw = gen.png_forward_filter(px, img_n, rng.integers(0, 5, y))
    color = {1: 0, 2: 4, 3: 2, 4: 6}[img_n]
    exp = O.png_create_image_raw(raw, img_n, out_n, x, y, 8, color)
    got = gpu_defilter(hip, raw, x, y, img_n, out_n, 8, color)[0]
    assert np.array_equal(got, exp)


@pytest.fixture(params=["host", "device", "device, pitched staging", "device, tight staging"])
def inflate_mode(request):
    """where the IDAT streams of the PNG batch call are inflated: zlib on the host threads, or k_inflate on the GPU (inflate.hip: batches of more
    files than host threads choose it themselves, GAMUT_HIP_PNG_INFLATE forces either) -- and, on the GPU, how the streams' slices go up: every
    stream `pitch` bytes apart and one hipMemcpy2DAsync per slice round, or a tight image and a copy per (stream, slice) (the call picks by how
    unequal the streams are; GAMUT_HIP_PNG_PITCHED forces either)"""
    keys = ("GAMUT_HIP_PNG_INFLATE", "GAMUT_HIP_PNG_PITCHED", "GAMUT_HIP_PNG_SLICE_KB")
    old = [os.environ.get(k) for k in keys]
    os.environ["GAMUT_HIP_PNG_INFLATE"] = request.param.split(",")[0]
    if "staging" in request.param:                               # 64 KiB slices: the larger files of the tests take several rounds
        os.environ["GAMUT_HIP_PNG_PITCHED"] = "1" if "pitched" in request.param else "0"
        os.environ["GAMUT_HIP_PNG_SLICE_KB"] = "64"
    yield request.param.split(",")[0]
    for k, v in zip(keys, old):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_png_file_batch_feeder(hip, inflate_mode):
    """gamut_hip_png_decode_batch_device: mixed files (sizes, colour types, depths, Adam7, palette + tRNS, a broken one) on a
    host thread pool + GPU == stbi_load / stbi_load_16 of the oracle, pixels left in HBM"""
    rng = np.random.default_rng(21)
    files = [open(os.path.join(HERE, "golden", "ref_images", n), "rb").read() for n in ("issue65.png", "vst3-compatible.png", "issue76.png")]
    w, h = 37, 23
    files += [gen.write_png(rng.integers(0, 256, (h, w * 3)), w, h, 2, 8),
              gen.write_png(rng.integers(0, 65536, (h, w * 4)), w, h, 6, 16),
              b"\x89PNG\r\n\x1a\n" + bytes(40),
              gen.write_png(rng.integers(0, 16, (h, w)), w, h, 3, 4, palette=rng.integers(0, 256, (16, 3)), trns=[0, 128, 255]),
              gen.write_png(rng.integers(0, 256, (h, w * 2)), w, h, 4, 8, interlace=1)]
    n = len(files)
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs])
    for req, bits in ((4, 8), (0, 8), (3, 16), (0, 0)):
        exp = []
        for f in files:
            try:
                to16 = bits == 16 or (bits == 0 and O.png_parse(f)["depth"] == 16)
                arr, comp = O.stbi_load(f, req, to16)
                exp.append(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))
            except Exception:
                exp.append(None)
        sizes = [e.size if e is not None else 0 for e in exp]
        offs = (np.concatenate([[0], np.cumsum(sizes)[:-1]]) + 0).astype(np.int64)
        dout = up(hip, np.full(int(sum(sizes)) + 64, 0xA5, np.uint8))
        info = (_capi.PngInfo * n)(); st = (C.c_int * n)()
        rc = hip.gamut_hip_png_decode_batch_device(ptrs, lens, n, req, bits, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, info, st, 3, None)
        host = down(hip, dout, int(sum(sizes)) + 64)
        hip.gamut_hip_device_free(dout)
        assert rc == _capi.ERR_DECODE and st[5] == _capi.ERR_DECODE and b"image 5" in hip.gamut_hip_last_error()
        for i, e in enumerate(exp):
            if e is None:
                assert i == 5
                continue
            assert st[i] == 0, (i, req, bits)
            assert np.array_equal(host[offs[i]:offs[i] + e.size], e), (i, req, bits)
            assert info[i].bits * info[i].channels * info[i].width * info[i].height // 8 == e.size
        assert (host[int(sum(sizes)):] == 0xA5).all()
    hd = _capi.PngInfo()
    assert hip.gamut_hip_png_read_header(ptrs[4], lens[4], C.byref(hd)) == 0 and (hd.width, hd.height, hd.bits, hd.channels) == (w, h, 16, 4)


def test_png_file_batch_same_geometry_groups(hip, inflate_mode):
    """files of one geometry are de-filtered in ONE launch through per-image offset tables (RGB8 -> rgba8: the alpha-inserting
    ring kernel; RGBA8 as is; grey + alpha insert through the scratch + expand route), next to files of another size"""
    rng = np.random.default_rng(33)
    w, h = 203, 131
    for color, ch, req in ((2, 3, 4), (6, 4, 4), (0, 1, 2), (2, 3, 3)):
        imgs = [rng.integers(0, 256, (h, w * ch)) for _ in range(5)]
        files = [gen.write_png(a, w, h, color, 8) for a in imgs] + [gen.write_png(rng.integers(0, 256, (9, 40 * ch)), 40, 9, color, 8)]
        files = files[:3] + files[5:] + files[3:5]                                   # the odd one in the middle
        n = len(files)
        bufs = [np.frombuffer(f, np.uint8) for f in files]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs])
        exp = [np.ascontiguousarray(O.stbi_load(f, req, False)[0]).view(np.uint8).reshape(-1) for f in files]
        for pad in (0, 1):                                                           # dword-aligned slots, then odd offsets
            sizes = [((e.size + 3) & ~3) + pad for e in exp]
            offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
            dout = up(hip, np.full(int(sum(sizes)) + 64, 0xA5, np.uint8))
            info = (_capi.PngInfo * n)(); st = (C.c_int * n)()
            _capi.check(hip.gamut_hip_png_decode_batch_device(ptrs, lens, n, req, 8, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, info, st, 4, None))
            host = down(hip, dout, int(sum(sizes)) + 64)
            hip.gamut_hip_device_free(dout)
            for i, e in enumerate(exp):
                assert st[i] == 0 and np.array_equal(host[offs[i]:offs[i] + e.size], e), (color, req, pad, i)
                assert (host[offs[i] + e.size:offs[i] + sizes[i]] == 0xA5).all(), "wrote past the image"


def test_files_written_by_libpng(hip):
    """real encoder output (Pillow / libpng: adaptive row filters, several zlib levels, palette, 1-bit, 16-bit grey) through
    the drop-in loader == the oracle == the pixels Pillow started from"""
    import io
    from PIL import Image
    rng = np.random.default_rng(31)
    w, h = 157, 83
    rgb = gen.synth_rgb(w, h, 91)
    alpha = ((np.add.outer(np.arange(h), np.arange(w)) * 3) % 256).astype(np.uint8)
    cases = [("RGB", rgb), ("RGBA", np.dstack([rgb, alpha])), ("L", rgb[:, :, 1]), ("LA", np.dstack([rgb[:, :, 1], alpha])),
             ("I;16", (rgb[:, :, 0].astype(np.uint16) * 257 + rgb[:, :, 1]).astype(np.uint16)), ("1", (rgb[:, :, 2] > 128))]
    for mode, arr in cases:
        im = Image.fromarray(arr) if mode != "1" else Image.fromarray(arr).convert("1")
        variants = [(im, dict(compress_level=1)), (im, dict(compress_level=9, optimize=True))]
        if mode == "RGB":
            variants.append((im.quantize(37), dict()))                   # palette image
        for img, kw in variants:
            bio = io.BytesIO(); img.save(bio, "PNG", **kw); data = bio.getvalue()
            buf = np.frombuffer(data, np.uint8)
            for req in (0, 4, 1):
                for sixteen in (False, True):
                    exp, comp = O.stbi_load(data, req, sixteen)
                    x, y, c = C.c_int(), C.c_int(), C.c_int(); f = C.c_float()
                    fn = hip.gamut_hip_stbi_load_16_from_memory if sixteen else hip.gamut_hip_stbi_load_from_memory
                    fn.restype = C.c_void_p
                    p = fn(buf.ctypes.data, buf.size, C.byref(x), C.byref(y), C.byref(c), req, C.byref(f), C.byref(f), C.byref(f))
                    assert p, (mode, hip.gamut_hip_last_error())
                    e8 = np.ascontiguousarray(exp).view(np.uint8).reshape(-1)
                    got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (e8.size,)).copy()
                    O._libc.free(C.c_void_p(p))
                    assert (x.value, y.value, c.value) == (w, h, comp) and np.array_equal(got, e8), (mode, req, sixteen)
            if img.mode in ("RGB", "RGBA", "L", "LA"):                     # and the oracle agrees with the source pixels
                src = np.asarray(img).reshape(h, w, -1)
                assert np.array_equal(O.stbi_load(data, 0, False)[0].reshape(h, w, -1), src), mode


def test_png_file_batch_of_libpng_files_inflates_on_the_device(hip):
    """48 files from a real encoder (Pillow / libpng, levels 1 / 6 / 9, RGB / RGBA / grey / 16-bit / palette / Adam7) in one
    batch on one host thread: large enough for the batch to pick the device inflate by itself (more files than threads); == the oracle's stbi_load per file.  One file has
    a damaged IDAT stream, one a bad zlib header: both are reported, the others decode."""
    import io
    from PIL import Image
    os.environ.pop("GAMUT_HIP_PNG_INFLATE", None)
    files = []
    for k in range(46):
        w, h = 61 + 17 * (k % 7), 40 + 11 * (k % 5)
        rgb = gen.synth_rgb(w, h, 300 + k)
        mode = k % 6
        if mode == 0: im = Image.fromarray(rgb)
        elif mode == 1: im = Image.fromarray(np.dstack([rgb, rgb[:, :, 0]]))
        elif mode == 2: im = Image.fromarray(rgb[:, :, 1])
        elif mode == 3: im = Image.fromarray((rgb[:, :, 0].astype(np.uint16) * 257 + rgb[:, :, 2]).astype(np.uint16))
        elif mode == 4: im = Image.fromarray(rgb).quantize(50)
        else: im = Image.fromarray(rgb)
        bio = io.BytesIO(); im.save(bio, "PNG", compress_level=(1, 6, 9)[k % 3], optimize=(k % 4 == 0))
        files.append(bio.getvalue())
    files[5] = gen.write_png(np.random.default_rng(2).integers(0, 256, (33, 47 * 3)), 47, 33, 2, 8, interlace=1)
    bad = bytearray(files[7]); i = bad.index(b"IDAT") + 4 + 40; bad[i] ^= 0x10; files.append(bytes(bad))             # damaged stream
    bad = bytearray(files[8]); i = bad.index(b"IDAT") + 4; bad[i] = 0x79; files.append(bytes(bad))                    # zlib header fails its check
    n = len(files)
    assert n >= 32
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs])
    exp = []
    for f in files:
        try:
            exp.append(np.ascontiguousarray(O.stbi_load(f, 4, False)[0]).view(np.uint8).reshape(-1))
        except Exception:
            exp.append(None)
    assert exp[-1] is None                                                        # (the damaged one may or may not still inflate)
    sizes = [e.size if e is not None else 0 for e in exp]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    dout = up(hip, np.full(int(sum(sizes)) + 64, 0xA5, np.uint8))
    info = (_capi.PngInfo * n)(); st = (C.c_int * n)()
    rc = hip.gamut_hip_png_decode_batch_device(ptrs, lens, n, 4, 8, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, info, st, 1, None)
    host = down(hip, dout, int(sum(sizes)) + 64)
    hip.gamut_hip_device_free(dout)
    assert rc == _capi.ERR_DECODE
    for i, e in enumerate(exp):
        if e is None:
            assert st[i] == _capi.ERR_DECODE, i
        else:
            assert st[i] == 0, (i, hip.gamut_hip_last_error())
            assert np.array_equal(host[offs[i]:offs[i] + e.size], e), i
    assert (host[int(sum(sizes)):] == 0xA5).all()
