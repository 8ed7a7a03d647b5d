"""The file-level batch entry points from several caller threads at once: each call owns a private copy stream, events, a
device arena and (PNG) workers with pooled pinned buffers -- results must equal those of the same calls made one by one."""
import ctypes as C
import fixtures
import os
import threading

import numpy as np
import pytest

import gen
from gamut_amd import _capi

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P64 = C.POINTER(C.c_int64)


def _down(L, p, n):
    out = np.empty(n, np.uint8)
    _capi.check(L.gamut_hip_memcpy_d2h(out.ctypes.data, p, n, None)); _capi.check(L.gamut_hip_stream_synchronize(None))
    return out


def _png_call(L, files, threads):
    n = len(files)
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[b.size for b in bufs])
    info = (_capi.PngInfo * n)()
    for i in range(n):
        _capi.check(L.gamut_hip_png_read_header(ptrs[i], lens[i], C.byref(info[i])))
    sizes = [((info[i].width * info[i].height * 4) + 3) & ~3 for i in range(n)]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    stream = L.gamut_hip_stream_create()
    dout = L.gamut_hip_device_malloc(int(sum(sizes)))
    st = (C.c_int * n)()
    _capi.check(L.gamut_hip_png_decode_batch_device(ptrs, lens, n, 4, 8, offs.ctypes.data_as(P64), dout, info, st, threads, stream))
    host = _down(L, dout, int(sum(sizes)))
    L.gamut_hip_device_free(dout); L.gamut_hip_stream_destroy(stream)
    return host


def _jpeg_call(L, files):
    k = len(files)
    bufs = [np.frombuffer(f, np.uint8) for f in files]
    ptrs = (C.c_void_p * k)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * k)(*[b.size for b in bufs])
    nblk = []
    for i in range(k):
        fr = _capi.JpegFrame()
        _capi.check(L.gamut_hip_jpeg_read_header(ptrs[i], lens[i], C.byref(fr)))
        nblk.append(fr.mcus_per_row * fr.mcus_per_col * fr.blocks_per_mcu)
    nblk = np.array(nblk, np.int64)
    co_off = np.concatenate([[0], np.cumsum(nblk * 64)[:-1]]).astype(np.int64); zz_off = np.concatenate([[0], np.cumsum(nblk)[:-1]]).astype(np.int64)
    stream = L.gamut_hip_stream_create()
    dco = L.gamut_hip_device_malloc(int(nblk.sum()) * 128); dzz = L.gamut_hip_device_malloc(int(nblk.sum()))
    info = (_capi.JpegFrame * k)(); st = (C.c_int * k)()
    _capi.check(L.gamut_hip_jpeg_entropy_decode_device(ptrs, lens, k, co_off.ctypes.data_as(P64), zz_off.ctypes.data_as(P64), dco, dzz, None, info, st, stream))
    host = np.concatenate([_down(L, dco, int(nblk.sum()) * 128), _down(L, dzz, int(nblk.sum()))])
    L.gamut_hip_device_free(dco); L.gamut_hip_device_free(dzz); L.gamut_hip_stream_destroy(stream)
    return host


def test_batch_entry_points_from_concurrent_threads(hip):
    rng = np.random.default_rng(41)
    w, h = 97, 71
    pngs = [gen.write_png(rng.integers(0, 256, (h, w * 3)), w, h, 2, 8) for _ in range(4)]                 # one geometry: a batched launch
    pngs += [gen.write_png(rng.integers(0, 256, (33, 50 * 4)), 50, 33, 6, 8), gen.write_png(rng.integers(0, 256, (h, w * 2)), w, h, 4, 8, interlace=1),
             open(os.path.join(G, "ref_images", "issue65.png"), "rb").read()]
    jpgs = []
    for p in fixtures.jpegs(issue35=False):
        d = open(p, "rb").read(); fr = _capi.JpegFrame()
        if not os.path.basename(p).startswith("p_") and hip.gamut_hip_jpeg_read_header(d, len(d), C.byref(fr)) == 0:
            jpgs.append(d)
    jpgs = jpgs * 3
    exp_png = _png_call(hip, pngs, 3); exp_jpg = _jpeg_call(hip, jpgs)
    errors = []

    def worker(tid):
        try:
            for rep in range(4):
                if (tid + rep) % 2:
                    if not np.array_equal(_png_call(hip, pngs, 2 + tid % 3), exp_png): errors.append((tid, rep, "png differs"))
                else:
                    if not np.array_equal(_jpeg_call(hip, jpgs), exp_jpg): errors.append((tid, rep, "jpeg differs"))
        except Exception as e:                # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors[:3]
