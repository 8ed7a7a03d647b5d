#!/usr/bin/env python3
"""Out-of-bounds probe for the device entry points (GPU AddressSanitizer is not available on this pool): every input and
output buffer of a call is placed so that it ENDS exactly at the end of its own hipMalloc allocation (a multiple of 2 MiB), so
a kernel that reads or writes past a buffer runs into unmapped memory and the process aborts with a memory access fault.
`python tools/oob_probe.py control` does such an over-read on purpose (exit code != 0 expected) to show the probe can see it.
Usage: python tools/oob_probe.py [control]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
from gamut_amd import _capi  # noqa: E402

L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
CHUNK = 2 << 20
VERBOSE = bool(os.environ.get('OOB_VERBOSE'))
allocs = []


def at_end(nbytes, align=1):
    """device pointer to nbytes whose last byte is the last byte of a fresh allocation"""
    total = (max(nbytes, 1) + CHUNK - 1) // CHUNK * CHUNK
    p = L.gamut_hip_device_malloc(total)
    assert p
    allocs.append(p)
    q = p + total - nbytes
    assert q % align == 0, (nbytes, align)
    return q


def up_end(arr, align=1):
    arr = np.ascontiguousarray(arr)
    q = at_end(arr.nbytes, align)
    _capi.check(L.gamut_hip_memcpy_h2d(q, arr.ctypes.data, arr.nbytes, None)); _capi.check(L.gamut_hip_stream_synchronize(None))
    return q


def free_all():
    _capi.check(L.gamut_hip_stream_synchronize(None))
    for p in allocs:
        L.gamut_hip_device_free(p)
    allocs.clear()


def png_cases():
    rng = np.random.default_rng(1)
    n = 0
    for img_n, depth, color in [(1, 1, 0), (1, 8, 0), (1, 16, 0), (2, 8, 4), (3, 8, 2), (3, 16, 2), (4, 8, 6), (4, 16, 6)]:
        fb = 1 if depth < 8 else img_n * (2 if depth == 16 else 1)
        for x, y in [(1, 1), (5, 2), (7, 5), (37, 1), (37, 3), (64, 65), (130, 66), (259, 71), (1000, 9), (33, 129)]:
            rows = gen.pack_samples(rng.integers(0, 1 << depth, (y, x * img_n)), depth)
            raw = gen.png_forward_filter(rows, fb, rng.integers(0, 5, y).astype(np.uint8))
            for out_n in ([img_n] + ([img_n + 1] if img_n in (1, 3) else [])):
                nbytes = x * y * out_n * (2 if depth == 16 else 1)
                if VERBOSE: print('png', img_n, depth, color, x, y, out_n, flush=True)
                draw = up_end(raw)
                dout = at_end(nbytes)
                dst = at_end(4, 4)
                _capi.check(L.gamut_hip_png_defilter_batch_device(draw, 0, raw.size, dout, nbytes, x, y, img_n, out_n, depth, color, 1, dst, None))
                free_all(); n += 1
    return n


def random_cases(n_png=250, n_jpeg=150):
    """seeded random geometries (widths and heights around the band / piece / MCU boundaries included)"""
    rng = np.random.default_rng(7)
    edge = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 191, 193]
    pick = lambda hi: int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, hi))
    fmts = [(1, 1, 0), (1, 2, 0), (1, 4, 0), (1, 8, 0), (1, 16, 0), (2, 8, 4), (2, 16, 4), (3, 8, 2), (3, 16, 2), (4, 8, 6), (4, 16, 6)]
    for _ in range(n_png):
        img_n, depth, color = fmts[int(rng.integers(0, len(fmts)))]
        x, y = pick(400), pick(200)
        fb = 1 if depth < 8 else img_n * (2 if depth == 16 else 1)
        out_n = img_n + (1 if img_n in (1, 3) and rng.random() < 0.5 else 0)
        rows = gen.pack_samples(rng.integers(0, 1 << depth, (y, x * img_n)), depth)
        raw = gen.png_forward_filter(rows, fb, rng.integers(0, 5, y).astype(np.uint8))
        nbytes = x * y * out_n * (2 if depth == 16 else 1)
        if VERBOSE: print('rpng', img_n, depth, color, x, y, out_n, flush=True)
        draw = up_end(raw); dout = at_end(nbytes); dst = at_end(4, 4)
        _capi.check(L.gamut_hip_png_defilter_batch_device(draw, 0, raw.size, dout, nbytes, x, y, img_n, out_n, depth, color, 1, dst, None))
        free_all()
    NB = {0: 1, 1: 3, 2: 4, 3: 4, 4: 6}; MCU = {0: (8, 8), 1: (8, 8), 2: (16, 8), 3: (8, 16), 4: (16, 16)}
    for _ in range(n_jpeg):
        st = int(rng.integers(0, 5)); oc = int(rng.choice([1, 3, 4]))
        w, h = pick(600), pick(150)
        mw, mh = MCU[st]
        nblk = ((w + mw - 1) // mw) * ((h + mh - 1) // mh) * NB[st]
        co = rng.integers(-300, 300, (nblk, 64)).astype(np.int16)
        if VERBOSE: print('rjpeg', st, w, h, oc, flush=True)
        dco = up_end(co, 16); dout = at_end(w * h * oc)
        _capi.check(L.gamut_hip_jpeg_reconstruct_batch_device(dco, nblk * 64, None, 0, dout, w * oc, w * h * oc, w, h, st, oc, 1, None))
        free_all()
    return n_png + n_jpeg


def jpeg_cases():
    rng = np.random.default_rng(2)
    NB = {0: 1, 1: 3, 2: 4, 3: 4, 4: 6}; MCU = {0: (8, 8), 1: (8, 8), 2: (16, 8), 3: (8, 16), 4: (16, 16)}
    n = 0
    for st in (0, 1, 2, 3, 4):
        for w, h in [(1, 1), (17, 9), (128, 16), (129, 17), (250, 33), (1920, 24)]:
            mw, mh = MCU[st]
            nblk = ((w + mw - 1) // mw) * ((h + mh - 1) // mh) * NB[st]
            co = rng.integers(-200, 200, (nblk, 64)).astype(np.int16)
            for oc in (4, 3, 1):
                if VERBOSE: print('jpeg', st, w, h, oc, flush=True)
                dco = up_end(co, 16)
                dout = at_end(w * h * oc)
                _capi.check(L.gamut_hip_jpeg_reconstruct_batch_device(dco, nblk * 64, None, 0, dout, w * oc, w * h * oc, w, h, st, oc, 1, None))
                free_all(); n += 1
    return n


def convert_cases(extra_rows=0):
    rng = np.random.default_rng(3)
    size = [1, 2, 4, 2, 4, 8, 2, 4, 8, 3, 6, 12, 4, 8, 16, 4, 8, 16]
    n = 0
    for s in range(18):
        for d in range(18):
            for w, h in [(1, 1), (7, 3), (64, 2), (1001, 3)]:
                src = rng.integers(0, 256, w * h * size[s]).astype(np.uint8)
                if size[s] in (4, 8, 12, 16) and s % 3 == 2:                    # f32 types: finite values
                    src = rng.random(w * h * size[s] // 4, dtype=np.float32).view(np.uint8)
                if VERBOSE: print('convert', s, d, w, h, flush=True)
                ds_ = up_end(src)
                dd = at_end(w * h * size[d])
                _capi.check(L.gamut_hip_scanlines_convert_device(s, ds_, w * size[s], 0, d, dd, w * size[d], 0, w, h + extra_rows, 1, None))
                free_all(); n += 1
            if extra_rows:
                return n
    return n



def inflate_cases():
    """DEFLATE streams that end with their allocation, outputs that end with theirs (capacity == size, capacity < size), at once and in
    slices (a sliced launch may read up to the bytes it was promised, never behind the stream)"""
    import zlib
    rng = np.random.default_rng(8)
    n = 0
    datas = [rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(), (rng.integers(0, 8, 900000, dtype=np.uint8) * 9).tobytes(), bytes(700000),
             b"abc" * 100000 + rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(), b"x"]
    for data in datas:
        for level in (0, 1, 6):
            co = zlib.compressobj(level, zlib.DEFLATED, -15); st = co.compress(data) + co.flush()
            for cap in (len(data), max(1, len(data) // 3)):
                for slice_bytes in (0, 40000):
                    dsrc = up_end(np.frombuffer(st, np.uint8)); ddst = at_end(cap); dlen = at_end(4, 4); dstat = at_end(4, 4)
                    desc = (_capi.InflateDesc * 1)()
                    desc[0].src = dsrc; desc[0].dst = ddst; desc[0].src_len = len(st); desc[0].dst_cap = cap
                    if slice_bytes: _capi.check(L.gamut_hip_inflate_batch_device_sliced(desc, 1, dlen, dstat, slice_bytes, None))
                    else: _capi.check(L.gamut_hip_inflate_batch_device(desc, 1, dlen, dstat, None))
                    _capi.check(L.gamut_hip_stream_synchronize(None))
                    got = np.empty(cap, np.uint8); _capi.check(L.gamut_hip_memcpy_d2h(got.ctypes.data, ddst, cap, None)); _capi.check(L.gamut_hip_stream_synchronize(None))
                    assert got.tobytes() == data[:cap], (len(data), level, cap, slice_bytes)
                    free_all(); n += 1
    return n


def batch_cases():
    """count > 1 with tight strides; JPEG with max_zag; QOI files resident in HBM; baseline files through the device entropy decoder"""
    import glob
    rng = np.random.default_rng(4)
    n = 0
    for img_n, out_n, x, y in [(4, 4, 61, 70), (3, 4, 37, 66), (3, 3, 64, 3), (1, 2, 130, 5)]:
        color = {1: 0, 3: 2, 4: 6}[img_n]
        raws = [gen.png_forward_filter(rng.integers(0, 256, (y, x * img_n), dtype=np.uint8), img_n, rng.integers(0, 5, y).astype(np.uint8)) for _ in range(5)]
        stride = raws[0].size
        draw = up_end(np.concatenate(raws)); dout = at_end(5 * x * y * out_n); dst = at_end(5 * 4, 4)
        _capi.check(L.gamut_hip_png_defilter_batch_device(draw, stride, stride, dout, x * y * out_n, x, y, img_n, out_n, 8, color, 5, dst, None))
        free_all(); n += 1
    for st, nb, (mw, mh) in ((4, 6, (16, 16)), (1, 3, (8, 8)), (0, 1, (8, 8))):
        for w, h in ((250, 33), (1920, 16)):
            nblk = ((w + mw - 1) // mw) * ((h + mh - 1) // mh) * nb
            co = rng.integers(-200, 200, (3, nblk, 64)).astype(np.int16)
            zz = rng.integers(1, 65, (3, nblk)).astype(np.uint8)
            for oc in (4, 3):
                dco = up_end(co, 16); dzz = up_end(zz); dout = at_end(3 * w * h * oc)
                _capi.check(L.gamut_hip_jpeg_reconstruct_batch_device(dco, nblk * 64, dzz, nblk, dout, w * oc, w * h * oc, w, h, st, oc, 3, None))
                free_all(); n += 1
    # QOI: the contract asks for GAMUT_HIP_QOI_SLACK readable bytes after every file -- and not one more
    files = [gen.qoi_encode(rng.integers(0, 256, (h, w, ch), dtype=np.uint8) // 64 * 64) for (w, h, ch) in ((33, 9, 3), (130, 70, 4), (5, 1, 3))]
    parts, begin, pos = [], [], 0
    for f in files:
        begin.append(pos); parts += [f, bytes(160)]; pos += len(f) + 160
    blob = np.frombuffer(b"".join(parts), np.uint8)
    dblob = up_end(blob)
    descs = (_capi.QoiDesc * len(files))()
    for i, f in enumerate(files):
        _capi.check(L.gamut_hip_qoi_read_header(f, len(f), C.byref(descs[i])))
    npx = [d.width * d.height for d in descs]
    offs = np.concatenate([[0], np.cumsum([p * 4 for p in npx])[:-1]]).astype(np.int64)
    dout = at_end(int(sum(npx)) * 4)
    b = np.array(begin, np.int64); sz = np.array([len(f) for f in files], np.int32)
    _capi.check(L.gamut_hip_qoi_decode_resident_device(dblob, blob.size, b.ctypes.data_as(C.POINTER(C.c_int64)), sz.ctypes.data_as(C.POINTER(C.c_int)), descs, len(files), 4,
                                                       offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, None))
    _capi.check(L.gamut_hip_stream_synchronize(None))
    free_all(); n += 1
    # baseline JPEG files: coefficients and max_zag of the whole batch end with their allocations
    jf = [open(p_, "rb").read() for p_ in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "jpeg", "*.jpg"))) if not os.path.basename(p_).startswith("p_")]
    frames = []
    for f in jf:
        fr = _capi.JpegFrame()
        if L.gamut_hip_jpeg_read_header(f, len(f), C.byref(fr)) == 0:
            frames.append((f, fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu))
    if frames:
        bufs = [np.frombuffer(f, np.uint8) for f, _ in frames]
        nblks = np.array([nb for _, nb in frames], np.int64)
        co_off = np.concatenate([[0], np.cumsum(nblks * 64)[:-1]]).astype(np.int64); zz_off = np.concatenate([[0], np.cumsum(nblks)[:-1]]).astype(np.int64)
        k = len(frames)
        ptrs = (C.c_void_p * k)(*[b_.ctypes.data for b_ in bufs]); lens = (C.c_size_t * k)(*[b_.size for b_ in bufs])
        dco = at_end(int(nblks.sum()) * 128, 16); dzz = at_end(int(nblks.sum()))
        info = (_capi.JpegFrame * k)(); st = (C.c_int * k)()
        L.gamut_hip_jpeg_entropy_decode_device(ptrs, lens, k, co_off.ctypes.data_as(C.POINTER(C.c_int64)), zz_off.ctypes.data_as(C.POINTER(C.c_int64)), dco, dzz, None, info, st, None)
        free_all(); n += 1
    return n


if len(sys.argv) > 1 and sys.argv[1] == "control":
    print("control: converting one row more than the buffers hold -- a memory access fault is the expected outcome", flush=True)
    L.gamut_hip_device_free(L.gamut_hip_device_malloc(64 << 20))
    n = 0
    rng = np.random.default_rng(3)
    w, h = 1 << 18, 8                                                     # 1 MiB rows: the extra rows are far outside the allocation
    src = rng.integers(0, 256, w * h * 4).astype(np.uint8)
    ds_ = up_end(src); dd = at_end(w * h * 4)
    _capi.check(L.gamut_hip_scanlines_convert_device(12, ds_, w * 4, 0, 12 + 3, dd, w * 4, 0, w, h + 64, 1, None))
    _capi.check(L.gamut_hip_stream_synchronize(None))
    print("control: NO fault (the probe is blind on this box)")
    sys.exit(0)

print("png", png_cases(), "cases ok", flush=True)
print("jpeg", jpeg_cases(), "cases ok", flush=True)
print("convert", convert_cases(), "cases ok", flush=True)
print("batch / qoi / entropy", batch_cases(), "cases ok", flush=True)
print("inflate", inflate_cases(), "cases ok", flush=True)
print("random geometries", random_cases(), "cases ok", flush=True)
print("oob_probe: no access outside any buffer")
