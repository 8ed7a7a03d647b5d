"""GPU parity: QOI decode (a workgroup of four waves per stream, or one wave per stream in large batches) through the C ABI vs the
CPU oracle; bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import gen
import oracle_lib as O
from gamut_amd import _capi
from test_oracle_pinning import _qoi_test_images

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["pipeline", "phases"])
def small_batch_kernel(request):
    """the two kernels for batches of fewer streams than the part has compute units (qoi.hip): k_qoi_pipe (wave 0 walks the groups
    while four producer waves prepare the next window and write the last one's pixels) and k_qoi_decode<4> (the same phases in turn)"""
    old = os.environ.get("GAMUT_HIP_QOI_PIPE")
    os.environ["GAMUT_HIP_QOI_PIPE"] = "1" if request.param == "pipeline" else "0"
    yield request.param
    if old is None:
        del os.environ["GAMUT_HIP_QOI_PIPE"]
    else:
        os.environ["GAMUT_HIP_QOI_PIPE"] = old


def test_qoi_drop_in_and_batch(hip, small_batch_kernel):
    imgs = _qoi_test_images()
    blobs = [gen.qoi_encode(a) for a in imgs]
    blobs.append(blobs[2][:200] + blobs[2][-8:])                       # ends early: the tail repeats the last pixel
    for data in blobs:
        buf = np.frombuffer(data, np.uint8)
        for ch in (0, 3, 4):
            exp, fc, cs = O.qoi_decode(data, ch)
            d = _capi.QoiDesc()
            p = hip.gamut_hip_qoi_decode(buf.ctypes.data, buf.size, C.byref(d), ch)
            assert p, hip.gamut_hip_last_error()
            got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (exp.size,)).copy()
            O._libc.free(C.c_void_p(p))
            assert (d.channels, d.colorspace, d.height) == (fc, cs, exp.shape[0])
            assert np.array_equal(got.reshape(exp.shape), exp)
    # batch entry point: mixed sizes in one launch, a bad file in the middle
    blobs.insert(2, b"qoif" + bytes(30))
    n = len(blobs)
    bufs = [np.frombuffer(b, np.uint8) for b in blobs]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); sizes = (C.c_int * n)(*[b.size for b in bufs])
    exp = [O.qoi_decode(b, 4) for b in blobs]
    nbytes = [e[0].size if e else 0 for e in exp]
    offs = np.concatenate([[0], np.cumsum(nbytes)[:-1]]).astype(np.int64)
    dout = hip.gamut_hip_device_malloc(int(sum(nbytes)) + 64)
    descs = (_capi.QoiDesc * n)(); st = (C.c_int * n)()
    rc = hip.gamut_hip_qoi_decode_batch_device(ptrs, sizes, n, 4, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, descs, st, None)
    assert rc == _capi.ERR_DECODE and st[2] == _capi.ERR_DECODE and [s for i, s in enumerate(st) if i != 2] == [0] * (n - 1)
    host = np.empty(int(sum(nbytes)), np.uint8)
    _capi.check(hip.gamut_hip_memcpy_d2h(host.ctypes.data, dout, host.nbytes, None))
    _capi.check(hip.gamut_hip_stream_synchronize(None))
    hip.gamut_hip_device_free(dout)
    for i, e in enumerate(exp):
        if e:
            assert np.array_equal(host[offs[i]:offs[i] + nbytes[i]], e[0].reshape(-1)), i


@pytest.mark.parametrize("mix", ["all", "some"])
def test_qoi_files_in_page_locked_memory(hip, mix):
    """Files that lie in page-locked host memory (gamut_hip_host_malloc_pinned) go up from where they are -- no staging copy -- beside pageable
    ones that are gathered first (qoi.hip, the upload of gamut_hip_qoi_decode_batch_device): every file pinned, and every other one, must give
    what the pageable call gives and what the oracle gives, a truncated, a damaged and a header-only file included; each pinned file ends
    exactly where its allocation's bytes end."""
    imgs = _qoi_test_images()
    blobs = [gen.qoi_encode(a) for a in imgs] * 3
    blobs.append(blobs[2][:200] + blobs[2][-8:])                       # ends early
    blobs.insert(4, b"qoif" + bytes(30))                               # a bad header in the middle
    damaged = bytearray(blobs[1]); damaged[40:48] = b"\xff" * 8; blobs.append(bytes(damaged))
    blobs.append(blobs[0][:14])                                        # the header alone
    n = len(blobs)
    hip.gamut_hip_host_malloc_pinned.restype = C.c_void_p; hip.gamut_hip_host_malloc_pinned.argtypes = [C.c_size_t]
    hip.gamut_hip_host_free_pinned.restype = None; hip.gamut_hip_host_free_pinned.argtypes = [C.c_void_p]
    exp = [O.qoi_decode(b, 4) for b in blobs]
    nbytes = [e[0].size if e else 0 for e in exp]
    offs = np.concatenate([[0], np.cumsum(nbytes)[:-1]]).astype(np.int64)

    def run(pin):
        keep, pinned, ptr_list = [], [], []
        for i, b in enumerate(blobs):
            if pin(i):
                q = hip.gamut_hip_host_malloc_pinned(len(b))
                assert q, hip.gamut_hip_last_error()
                C.memmove(q, b, len(b)); pinned.append(q); ptr_list.append(q)
            else:
                a = np.frombuffer(b, np.uint8); keep.append(a); ptr_list.append(a.ctypes.data)
        ptrs = (C.c_void_p * n)(*ptr_list); sizes = (C.c_int * n)(*[len(b) for b in blobs])
        dout = hip.gamut_hip_device_malloc(int(sum(nbytes)) + 64)
        descs = (_capi.QoiDesc * n)(); st = (C.c_int * n)()
        rc = hip.gamut_hip_qoi_decode_batch_device(ptrs, sizes, n, 4, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, descs, st, None)
        host = np.empty(int(sum(nbytes)), np.uint8)
        _capi.check(hip.gamut_hip_memcpy_d2h(host.ctypes.data, dout, host.nbytes, None))
        _capi.check(hip.gamut_hip_stream_synchronize(None))
        hip.gamut_hip_device_free(dout)
        for q in pinned:
            hip.gamut_hip_host_free_pinned(q)
        return rc, list(st), host

    rc0, st0, host0 = run(lambda i: False)
    rc1, st1, host1 = run((lambda i: True) if mix == "all" else (lambda i: i % 2 == 1))
    assert (rc1, st1) == (rc0, st0) and rc0 == _capi.ERR_DECODE
    assert [s != 0 for s in st0] == [e is None for e in exp]
    for i, e in enumerate(exp):
        if e:
            assert np.array_equal(host0[offs[i]:offs[i] + nbytes[i]], e[0].reshape(-1)), i
            assert np.array_equal(host1[offs[i]:offs[i] + nbytes[i]], e[0].reshape(-1)), i


def test_qoi_resident_streams(hip, small_batch_kernel):
    """files already in HBM (gamut_hip_qoi_decode_resident_device): same pixels as the host-pointer batch; bounds are checked"""
    from gamut_amd import synth
    rng = np.random.default_rng(5)
    imgs = _qoi_test_images()[2:] + [rng.integers(0, 256, (40, 300, 3), dtype=np.uint8), np.zeros((70, 90, 4), np.uint8)]
    files = [synth.qoi_encode(a) for a in imgs]
    assert files[0] == gen.qoi_encode(imgs[0])                       # the vectorised encoder = the specification encoder
    n = len(files)
    begin, pos, parts = [], 3, [bytes(3)]                            # an odd base offset: streams have no alignment
    for f in files:
        begin.append(pos); parts += [f, bytes(160)]; pos += len(f) + 160
    hblob = np.frombuffer(b"".join(parts), np.uint8).copy()
    blob = hip.gamut_hip_device_malloc(hblob.size)
    _capi.check(hip.gamut_hip_memcpy_h2d(blob, hblob.ctypes.data, hblob.size, None))
    descs = (_capi.QoiDesc * n)()
    for i, f in enumerate(files):
        _capi.check(hip.gamut_hip_qoi_read_header(f, len(f), C.byref(descs[i])))
    nbytes = [a.shape[0] * a.shape[1] * 4 for a in imgs]
    offs = np.concatenate([[0], np.cumsum(nbytes)[:-1]]).astype(np.int64)
    out = hip.gamut_hip_device_malloc(int(sum(nbytes)))
    b = np.array(begin, np.int64); sz = np.array([len(f) for f in files], np.int32)
    args = lambda blob_len: (blob, blob_len, b.ctypes.data_as(C.POINTER(C.c_int64)), sz.ctypes.data_as(C.POINTER(C.c_int)), descs, n, 4,
                             offs.ctypes.data_as(C.POINTER(C.c_int64)), out, None)
    _capi.check(hip.gamut_hip_stream_synchronize(None))
    _capi.check(hip.gamut_hip_qoi_decode_resident_device(*args(hblob.size)))
    host = np.empty(int(sum(nbytes)), np.uint8)
    _capi.check(hip.gamut_hip_memcpy_d2h(host.ctypes.data, out, host.nbytes, None))
    _capi.check(hip.gamut_hip_stream_synchronize(None))
    for i, f in enumerate(files):
        exp = O.qoi_decode(f, 4)[0].reshape(-1)
        assert np.array_equal(host[offs[i]:offs[i] + nbytes[i]], exp), i
    assert hip.gamut_hip_qoi_decode_resident_device(*args(hblob.size - 1)) == _capi.ERR_INVALID_ARG      # last stream's slack is cut
    assert hip.gamut_hip_qoi_decode_resident_device(*args(hblob.size)[:6], 5, *args(0)[7:]) == _capi.ERR_INVALID_ARG
    hip.gamut_hip_device_free(blob); hip.gamut_hip_device_free(out)


def _arbitrary_streams():
    """streams no encoder writes: random chunk bytes under a valid header (every byte sequence is a QOI op sequence) -- too short, too
    long, op mixes that change every window's entry offset, runs across the end of the image"""
    rng = np.random.default_rng(31)
    out = []
    for k in range(24):
        w, h, ch = int(rng.integers(1, 400)), int(rng.integers(1, 60)), 3 + (k & 1)
        need = w * h
        nbytes = int(rng.integers(0, 3 * need + 40))
        body = rng.integers(0, 256, nbytes, dtype=np.uint8)
        if k % 3 == 0:
            body[rng.random(nbytes) < 0.5] = 0xC0 | int(rng.integers(0, 62))          # many RUN ops
        if k % 4 == 1:
            body[rng.random(nbytes) < 0.6] &= 0x3F                                    # many INDEX ops
        out.append(b"qoif" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([ch, 0]) + body.tobytes() + bytes(7) + b"\x01")
    # RGB, RGBA and INDEX ops close together among DIFF / LUMA ops: every combination of "the most recent op that set the colour" and "the most
    # recent op that set alpha" inside one group of 64 ops (qoi_group_scan's three alpha cases), in densities from a few per group to most ops
    for k in range(16):
        w, h, ch = int(rng.integers(40, 500)), int(rng.integers(8, 50)), 3 + (k & 1)
        nbytes = int(rng.integers(w * h, 3 * w * h))
        body = rng.integers(0x40, 0xC0, nbytes, dtype=np.uint8)                         # DIFF and LUMA ops (a LUMA op's second byte: any of these)
        p_rgb, p_rgba, p_idx = [(0.02, 0.0, 0.01), (0.0, 0.02, 0.02), (0.05, 0.05, 0.05), (0.3, 0.2, 0.2)][k // 4]
        r = rng.random(nbytes)
        body[r < p_rgb] = 0xFE
        body[(r >= p_rgb) & (r < p_rgb + p_rgba)] = 0xFF
        body[(r >= p_rgb + p_rgba) & (r < p_rgb + p_rgba + p_idx)] &= 0x3F
        if k % 4 == 3:
            body[rng.random(nbytes) < 0.03] = 0xC0 | int(rng.integers(0, 62))
        out.append(b"qoif" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([ch, 0]) + body.tobytes() + bytes(7) + b"\x01")
    return out


def test_qoi_arbitrary_streams(hip, small_batch_kernel):
    files = _arbitrary_streams()
    for reps in (1, 24):                                       # 40 streams: the small-batch kernels; 960: one wave per stream
        blobs = files * reps
        n = len(blobs)
        if reps > 1 and small_batch_kernel == "phases":
            continue
        bufs = [np.frombuffer(b, np.uint8) for b in blobs]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); sizes = (C.c_int * n)(*[b.size for b in bufs])
        exp = [O.qoi_decode(b, 4)[0].reshape(-1) for b in files] * reps
        nbytes = [e.size for e in exp]
        offs = np.concatenate([[0], np.cumsum(nbytes)[:-1]]).astype(np.int64)
        dout = hip.gamut_hip_device_malloc(int(sum(nbytes)) + 64)
        descs = (_capi.QoiDesc * n)(); st = (C.c_int * n)()
        _capi.check(hip.gamut_hip_qoi_decode_batch_device(ptrs, sizes, n, 4, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, descs, st, None))
        host = np.empty(int(sum(nbytes)), np.uint8)
        _capi.check(hip.gamut_hip_memcpy_d2h(host.ctypes.data, dout, host.nbytes, None))
        _capi.check(hip.gamut_hip_stream_synchronize(None))
        hip.gamut_hip_device_free(dout)
        for i in range(n):
            assert np.array_equal(host[offs[i]:offs[i] + nbytes[i]], exp[i]), (reps, i % len(files))


def test_qoi_large_batches(hip):
    """800 streams in one call.  From host memory (gamut_hip_qoi_decode_batch_device) the files go up in groups of 256 on a copy stream and
    every group is decoded behind its own upload; resident in HBM (gamut_hip_qoi_decode_resident_device) >= 768 streams run one wave per
    stream (k_qoi_decode<1>).  Same pixels as the oracle either way, rgba and rgb outputs."""
    rng = np.random.default_rng(9)
    imgs = []
    for k in range(16):
        w, h = 30 + 7 * k, 20 + 3 * (k % 5)
        a = rng.integers(0, 256, (h, w, 3 + (k % 2)), dtype=np.uint8)
        a[h // 3: h // 2] = a[h // 3: h // 3 + 1, :1]                            # runs
        a[:, w // 2:] = a[:, : w - w // 2] // 16 * 16                            # few colours: INDEX ops
        imgs.append(a)
    files = [gen.qoi_encode(a) for a in imgs]
    n = 800
    for ch in (4, 3):
        exp = [O.qoi_decode(f, ch)[0].reshape(-1) for f in files]
        pick = [i % len(files) for i in range(n)]
        bufs = [np.frombuffer(files[i], np.uint8) for i in pick]
        ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); sizes = (C.c_int * n)(*[b.size for b in bufs])
        nbytes = [exp[i].size for i in pick]
        offs = np.concatenate([[0], np.cumsum(nbytes)[:-1]]).astype(np.int64)
        dout = hip.gamut_hip_device_malloc(int(sum(nbytes)) + 64)
        descs = (_capi.QoiDesc * n)(); st = (C.c_int * n)()
        host = np.empty(int(sum(nbytes)), np.uint8)
        for resident in (False, True):
            _capi.check(hip.gamut_hip_memcpy_h2d(dout, np.full(host.size, 0x5A, np.uint8).ctypes.data, host.size, None))
            if not resident:
                _capi.check(hip.gamut_hip_qoi_decode_batch_device(ptrs, sizes, n, ch, offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, descs, st, None))
            else:
                begin, pos, parts = [], 0, []
                for i in pick:
                    begin.append(pos); parts += [files[i], bytes(160)]; pos += len(files[i]) + 160
                hblob = np.frombuffer(b"".join(parts), np.uint8).copy()
                blob = hip.gamut_hip_device_malloc(hblob.size)
                _capi.check(hip.gamut_hip_memcpy_h2d(blob, hblob.ctypes.data, hblob.size, None))
                b = np.array(begin, np.int64); sz = np.array([len(files[i]) for i in pick], np.int32)
                _capi.check(hip.gamut_hip_stream_synchronize(None))
                _capi.check(hip.gamut_hip_qoi_decode_resident_device(blob, hblob.size, b.ctypes.data_as(C.POINTER(C.c_int64)), sz.ctypes.data_as(C.POINTER(C.c_int)), descs, n, ch,
                                                                      offs.ctypes.data_as(C.POINTER(C.c_int64)), dout, None))
            _capi.check(hip.gamut_hip_memcpy_d2h(host.ctypes.data, dout, host.nbytes, None))
            _capi.check(hip.gamut_hip_stream_synchronize(None))
            if resident:
                hip.gamut_hip_device_free(blob)
            for k in range(n):
                assert np.array_equal(host[offs[k]:offs[k] + nbytes[k]], exp[pick[k]]), (ch, resident, k)
        hip.gamut_hip_device_free(dout)


def test_image_load_qoi(hip):
    """Image.loadFromMemory on a QOI file (plugins/qoi.d:47-141): rgb8 / rgba8 as in the file, then convertTo per load flags"""
    from gamut_amd.image import Image
    for a in _qoi_test_images()[2:4]:
        h, w, ch = a.shape
        data = gen.qoi_encode(a)
        im = Image()
        assert im.loadFromMemory(data), im.errorMessage()
        assert (im.width, im.height) == (w, h) and im.type == (O.PT["rgb8"] if ch == 3 else O.PT["rgba8"])
        assert np.array_equal(im.pixels().reshape(h, w, ch), a)
