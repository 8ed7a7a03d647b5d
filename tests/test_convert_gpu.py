"""GPU parity: the HIP scanline conversion matrix (through the C ABI) vs the CPU oracle.

Bar: bit-exact for every PixelType pair -- integer outputs byte-for-byte, f32
outputs compared as raw 32-bit patterns (0 ulp vs IEEE per-operation binary32;
the stated tolerance against "whatever D compiler built the reference" is 1 ulp,
see DESIGN.md).
"""
import ctypes as C

import numpy as np
import pytest

import gen
import oracle_lib as O
from oracle_lib import PIXEL_TYPES, PT, PT_SIZE

pytestmark = pytest.mark.gpu


class DevBuf:
    def __init__(self, L, host):
        self.L, self.n = L, host.size
        self.p = L.gamut_hip_device_malloc(max(1, host.size))
        assert self.p
        from gamut_amd import _capi
        _capi.check(L.gamut_hip_memcpy_h2d(self.p, host.ctypes.data, host.size, None))
        _capi.check(L.gamut_hip_stream_synchronize(None))

    def get(self):
        from gamut_amd import _capi
        out = np.empty(self.n, np.uint8)
        _capi.check(self.L.gamut_hip_stream_synchronize(None))
        _capi.check(self.L.gamut_hip_memcpy_d2h(out.ctypes.data, self.p, self.n, None))
        _capi.check(self.L.gamut_hip_stream_synchronize(None))
        return out

    def free(self):
        self.L.gamut_hip_device_free(self.p)


def _run_device(L, st, dt, sbuf, soff, spitch, dbuf_size, doff, dpitch, w, h, layers=1, slayer=0, dlayer=0):
    from gamut_amd import _capi
    ds = DevBuf(L, sbuf)
    dd = DevBuf(L, np.full(dbuf_size, 0xA5, np.uint8))
    _capi.check(L.gamut_hip_scanlines_convert_device(st, ds.p + soff, spitch, slayer, dt, dd.p + doff, dpitch, dlayer,
                                                     w, h, layers, None))
    out = dd.get()
    ds.free(); dd.free()
    return out


@pytest.mark.parametrize("src", PIXEL_TYPES)
def test_all_pairs_device_bit_exact(hip, src):
    """every destination type from `src`, three layouts: gapless/aligned, padded unaligned pitch, flipped."""
    rng = np.random.default_rng(1234 + PT[src])
    st = PT[src]
    for dst in PIXEL_TYPES:
        dt = PT[dst]
        for (w, h, spad, dpad, flip) in [(67, 9, 0, 0, False), (33, 5, 3, 5, False), (21, 4, 16, 32, True), (256, 2, 0, 0, False)]:
            px = gen.make_pixels(src, w * h, rng)
            spitch, dpitch = w * PT_SIZE[st] + spad, w * PT_SIZE[dt] + dpad
            sbuf, soff, sp = gen.pack_rows(px, w, h, spitch, flipped=flip)
            # expected (oracle) into a 0xA5-filled buffer of the same geometry
            exp = np.full(dpitch * h + 32, 0xA5, np.uint8)
            doff, dp = ((h - 1) * dpitch, -dpitch) if flip else (0, dpitch)
            inter = O.lib().orc_scanlines_inter_type(st, dt)
            ibuf = np.zeros(w * 16 + 16, np.uint8)
            assert O.lib().orc_scanlines_convert(st, sbuf.ctypes.data + soff, sp, dt, exp.ctypes.data + doff, dp, w, h,
                                                 inter, ibuf.ctypes.data)
            got = _run_device(hip, st, dt, sbuf, soff, sp, exp.size, doff, dp, w, h)
            if not np.array_equal(got, exp):
                bad = np.flatnonzero(got != exp)
                raise AssertionError(f"{src}->{dst} w={w} h={h} flip={flip}: {bad.size} bytes differ, first at {bad[:8]}; "
                                     f"got {got[bad[:8]]} exp {exp[bad[:8]]}")


def test_layers_and_gapless_flatten(hip):
    """layered buffers (Image.convertTo's layer loop, image.d:1273-1311), contiguous and with a gap between layers."""
    rng = np.random.default_rng(7)
    w, h, layers = 50, 7, 3
    for src, dst in [("rgba16", "rgbaf32"), ("rgbaf32", "rgba8"), ("rgba8", "rgba16"), ("rgb8", "l16"), ("la8", "rgb8")]:
        st, dt = PT[src], PT[dst]
        for gap in (0, 64):
            sl, dl = w * h * PT_SIZE[st] + gap, w * h * PT_SIZE[dt] + gap
            px = gen.make_pixels(src, w * h * layers, rng).reshape(layers, h * w, -1)
            sbuf = np.full(sl * layers + 16, 0xA5, np.uint8)
            exp = np.full(dl * layers + 16, 0xA5, np.uint8)
            for l in range(layers):
                row = px[l].view(np.uint8).reshape(-1)
                sbuf[l * sl: l * sl + row.size] = row
                e = O.scanlines_convert(st, row, dt, w, h)
                exp[l * dl: l * dl + e.size] = e
            got = _run_device(hip, st, dt, sbuf, 0, w * PT_SIZE[st], exp.size, 0, w * PT_SIZE[dt], w, h, layers, sl, dl)
            assert np.array_equal(got, exp), f"{src}->{dst} gap={gap}"


def test_exhaustive_integer_tables(hip):
    """all 256 / 65536 codes for the u8/u16 <-> f32 and u8 <-> u16 composites (SURVEY.md 8a note)."""
    u8 = np.arange(256, dtype=np.uint8).repeat(4).reshape(-1, 4)
    u16 = np.arange(65536, dtype=np.uint16).repeat(4).reshape(-1, 4)
    for src, arr, dsts in [("rgba8", u8, ["rgbaf32", "rgba16", "rgbap16", "l16"]),
                           ("rgba16", u16, ["rgbaf32", "rgba8", "rgbap8", "la8"])]:
        st = PT[src]
        n = arr.shape[0]
        for dst in dsts:
            dt = PT[dst]
            exp = O.scanlines_convert(st, arr, dt, n, 1)
            got = _run_device(hip, st, dt, arr.view(np.uint8).reshape(-1).copy(), 0, n * PT_SIZE[st], exp.size, 0,
                              n * PT_SIZE[dt], n, 1)
            assert np.array_equal(got, exp), f"{src}->{dst}"
    # the composites the reference ends up computing (verified in SURVEY.md): v*257 and (v*255+32767)//65535
    got = _run_device(hip, PT["rgba8"], PT["rgba16"], u8.reshape(-1).copy(), 0, 1024, 2048, 0, 2048, 256, 1).view(np.uint16)
    assert np.array_equal(got, (u8.reshape(-1).astype(np.uint32) * 257).astype(np.uint16))
    got = _run_device(hip, PT["rgba16"], PT["rgba8"], u16.view(np.uint8).reshape(-1).copy(), 0, 65536 * 8, 65536 * 4, 0,
                      65536 * 4, 65536, 1)
    assert np.array_equal(got, ((u16.reshape(-1).astype(np.uint64) * 255 + 32767) // 65535).astype(np.uint8))


def test_every_8bit_premultiplied_quotient(hip):
    """8-bit premultiplied sources divide colour by alpha with one reciprocal per pixel and a Markstein correction per channel
    (convert.hip: div_by_alpha8) instead of an IEEE division: every (colour, alpha) code pair -- 65 536 per source type, alpha 0
    and colour > alpha included -- must give the oracle's bits, seen through the f32 destination and the integer ones."""
    i, j = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    i, j = i.reshape(-1), j.reshape(-1)
    n = i.size
    cases = [("rgbap8", np.stack([i, (255 - i).astype(np.uint8), (i * 3 + 1).astype(np.uint8), j], axis=1), ["rgbaf32", "rgbf32", "rgba8", "rgba16", "l16", "laf32", "rgbap16"]),
             ("lap8", np.stack([i, j], axis=1), ["laf32", "rgbaf32", "la8", "l16", "rgb16", "lapf32"])]
    for src, arr, dsts in cases:
        st = PT[src]
        for dst in dsts:
            dt = PT[dst]
            exp = O.scanlines_convert(st, arr, dt, n, 1)
            got = _run_device(hip, st, dt, np.ascontiguousarray(arr).reshape(-1).copy(), 0, n * PT_SIZE[st], exp.size, 0, n * PT_SIZE[dt], n, 1)
            assert np.array_equal(got, exp), f"{src}->{dst}: {np.count_nonzero(got != exp)} bytes differ"


def test_f32_boundaries(hip):
    """f32 -> u8/u16 at every half-integer boundary +-1 ulp, out-of-range and non-finite-free extremes."""
    e = gen.f32_edge_values()
    n = (e.size // 4) * 4
    px = e[:n].reshape(-1, 4).copy()
    st = PT["rgbaf32"]
    for dst in ["rgba8", "rgba16", "rgb8", "rgb16", "l8", "l16", "la8", "la16", "lap8", "lap16", "rgbap8", "rgbap16",
                "lf32", "laf32", "lapf32", "rgbf32", "rgbapf32"]:
        dt = PT[dst]
        w = px.shape[0]
        exp = O.scanlines_convert(st, px, dt, w, 1)
        got = _run_device(hip, st, dt, px.view(np.uint8).reshape(-1).copy(), 0, w * 16, exp.size, 0, w * PT_SIZE[dt], w, 1)
        assert np.array_equal(got, exp), dst


def test_host_dropin_matches_scanlinesConvert(hip):
    """gamut_hip_scanlines_convert: host pointers, signed pitches, gap bytes untouched (scanline.d:70-121)."""
    from gamut_amd import _capi
    rng = np.random.default_rng(99)
    for src, dst, flip_s, flip_d in [("rgb8", "rgbaf32", False, True), ("rgbaf32", "rgba8", True, False),
                                     ("l8", "rgba8", False, False), ("rgba16", "rgba16", True, False),
                                     ("rgbap16", "la8", False, False)]:
        st, dt = PT[src], PT[dst]
        w, h = 45, 6
        px = gen.make_pixels(src, w * h, rng)
        spitch, dpitch = w * PT_SIZE[st] + 7, w * PT_SIZE[dt] + 9
        sbuf, soff, sp = gen.pack_rows(px, w, h, spitch, flipped=flip_s)
        exp = np.full(dpitch * h + 32, 0xA5, np.uint8)
        got = exp.copy()
        doff, dp = ((h - 1) * dpitch, -dpitch) if flip_d else (0, dpitch)
        inter = O.lib().orc_scanlines_inter_type(st, dt)
        ibuf = np.zeros(w * 16 + 16, np.uint8)
        assert O.lib().orc_scanlines_convert(st, sbuf.ctypes.data + soff, sp, dt, exp.ctypes.data + doff, dp, w, h, inter,
                                             ibuf.ctypes.data)
        _capi.check(hip.gamut_hip_scanlines_convert(st, sbuf.ctypes.data + soff, sp, dt, got.ctypes.data + doff, dp, w, h))
        assert np.array_equal(got, exp), f"{src}->{dst}"
    # zero-size and invalid arguments follow the reference's conventions
    assert hip.gamut_hip_scanlines_convert(12, None, 0, 14, None, 0, 0, 0) == 0
    assert hip.gamut_hip_scanlines_convert(-1, sbuf.ctypes.data, 4, 14, got.ctypes.data, 16, 1, 1) == _capi.ERR_INVALID_ARG
    assert b"PixelType" in hip.gamut_hip_last_error()


def test_every_16bit_and_8bit_value_decodes_exactly(hip):
    """the kernel divides by 255 / 65535 with an FMA sequence instead of a division: every input value, bit for bit."""
    for src, n, dt in (("l16", 65536, np.uint16), ("l8", 256, np.uint8)):
        px = np.arange(n, dtype=dt).view(np.uint8)
        for dst in ("rgbaf32", "lf32", "rgba16", "rgba8"):
            exp = O.scanlines_convert(PT[src], px, PT[dst], n, 1)
            got = _run_device(hip, PT[src], PT[dst], px, 0, px.size, exp.size, 0, exp.size, n, 1)
            assert np.array_equal(got, exp), (src, dst)


def test_full_size_round_trip_properties(hip):
    """BASELINE.json config 4 geometry (8192 x 8192), checked through size-independent properties: rgba8 -> rgba16 -> rgba8 and
    rgba8 -> rgbaf32 -> rgba8 are the identity (v*257, then (v*255+32767)/65535; v/255.0f, then (int)(0.5f + f*255.0f)), and
    rgba16 -> rgbaf32 -> rgba16 is the identity on every 16-bit value."""
    from gamut_amd import _capi
    L = hip
    w = h = 8192
    rng = np.random.default_rng(12)
    src = rng.integers(0, 256, w * h * 4, dtype=np.uint8)
    a = DevBuf(L, src)
    b = DevBuf(L, np.zeros(16, np.uint8)); L.gamut_hip_device_free(b.p); b.p = L.gamut_hip_device_malloc(w * h * 16); b.n = w * h * 16
    c = DevBuf(L, np.zeros(16, np.uint8)); L.gamut_hip_device_free(c.p); c.p = L.gamut_hip_device_malloc(w * h * 4); c.n = w * h * 4
    for mid, msz in (("rgba16", 8), ("rgbaf32", 16)):
        _capi.check(L.gamut_hip_scanlines_convert_device(PT["rgba8"], a.p, w * 4, 0, PT[mid], b.p, w * msz, 0, w, h, 1, None))
        _capi.check(L.gamut_hip_scanlines_convert_device(PT[mid], b.p, w * msz, 0, PT["rgba8"], c.p, w * 4, 0, w, h, 1, None))
        assert np.array_equal(c.get(), src), mid
    # 16-bit: every value appears (w*h*4 samples of a repeating 0..65535 ramp)
    ramp = np.tile(np.arange(65536, dtype=np.uint16), (w * h * 4) // 65536).view(np.uint8)
    d = DevBuf(L, ramp)
    e = DevBuf(L, np.zeros(16, np.uint8)); L.gamut_hip_device_free(e.p); e.p = L.gamut_hip_device_malloc(w * h * 8); e.n = w * h * 8
    _capi.check(L.gamut_hip_scanlines_convert_device(PT["rgba16"], d.p, w * 8, 0, PT["rgbaf32"], b.p, w * 16, 0, w, h, 1, None))
    _capi.check(L.gamut_hip_scanlines_convert_device(PT["rgbaf32"], b.p, w * 16, 0, PT["rgba16"], e.p, w * 8, 0, w, h, 1, None))
    assert np.array_equal(e.get(), ramp)
    for buf in (a, b, c, d, e):
        buf.free()


@pytest.mark.parametrize("w,h", [(4096, 4096), (4099, 4093)])
def test_more_gapless_layers_than_one_launch_indexes(hip, w, h):
    """257 gapless layers of 4096 x 4096 l8 <-> rgbaf32 are 4.31 G units -- more than a launch indexes with 32 bits, so the library converts them as
    several launches of whole layers (launch_pair, convert.hip).  Rows of the first layer, of the layers either side of the cut and of the last layer
    against the oracle, both directions; 77 GB of HBM.  4099 x 4093: layers of an odd number of bytes -- the later launches start at addresses that are
    no multiple of 16 (the byte-wise kernel) and every launch has a tail of its own."""
    from gamut_amd import _capi
    L = hip
    layers = 257
    npx = w * h
    assert npx * layers >= 0xFFFFFFF0
    rng = np.random.default_rng(77)
    base = rng.integers(0, 256, npx, dtype=np.uint8)
    src = L.gamut_hip_device_malloc(npx * layers); mid = L.gamut_hip_device_malloc(npx * layers * 16); back = L.gamut_hip_device_malloc(npx * layers)
    assert src and mid and back
    try:
        for k in range(layers):
            layer = base ^ np.uint8(k & 255)
            _capi.check(L.gamut_hip_memcpy_h2d(src + k * npx, layer.ctypes.data, npx, None))
            _capi.check(L.gamut_hip_stream_synchronize(None))          # (the host array is reused)
        _capi.check(L.gamut_hip_scanlines_convert_device(PT["l8"], src, w, w * h, PT["rgbaf32"], mid, w * 16, w * h * 16, w, h, layers, None))
        _capi.check(L.gamut_hip_scanlines_convert_device(PT["rgbaf32"], mid, w * 16, w * h * 16, PT["l8"], back, w, w * h, w, h, layers, None))
        _capi.check(L.gamut_hip_stream_synchronize(None))
        per = 0xFFFFFFF0 // (npx + 1) - 1                                # layers per launch (G = 1 for this pair)
        assert 1 <= per < layers
        rows = 8
        for k in sorted({0, per - 1, per, layers - 1}):
            for r0 in (0, h // 2 + 5, h - rows):
                a = (base ^ np.uint8(k & 255))[r0 * w:(r0 + rows) * w]
                f = np.empty(rows * w * 16, np.uint8); b = np.empty(rows * w, np.uint8)
                _capi.check(L.gamut_hip_memcpy_d2h(f.ctypes.data, mid + (k * npx + r0 * w) * 16, f.size, None))
                _capi.check(L.gamut_hip_memcpy_d2h(b.ctypes.data, back + k * npx + r0 * w, b.size, None))
                _capi.check(L.gamut_hip_stream_synchronize(None))
                exp_f = O.scanlines_convert(PT["l8"], a, PT["rgbaf32"], w, rows)
                assert np.array_equal(f, exp_f), (k, r0)
                assert np.array_equal(b, O.scanlines_convert(PT["rgbaf32"], exp_f, PT["l8"], w, rows)), (k, r0)
    finally:
        for p in (src, mid, back):
            L.gamut_hip_device_free(p)
