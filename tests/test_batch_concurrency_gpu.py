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


def _mixed_files():
    """JPEG (baseline + progressive + restart markers), PNG (RGBA8, palette, 16-bit, Adam7), QOI (rgb, rgba) in one list"""
    import io
    from PIL import Image
    import oracle_lib as O
    files, exp = [], []
    rng = np.random.default_rng(11)
    for p in fixtures.jpegs()[:6]:
        b = open(p, "rb").read()
        files.append(b); exp.append(lambda rc, b=b: O.decompress_jpeg(b, rc)[0].reshape(-1))
    L = _capi.lib()
    for p in fixtures.ref_pngs():
        b = open(p, "rb").read()
        hdr = _capi.PngInfo(); buf = np.frombuffer(b, np.uint8)
        if L.gamut_hip_png_read_header(buf.ctypes.data, buf.size, C.byref(hdr)) != 0 or hdr.width * hdr.height * 4 > (1 << 22):
            continue                                            # (the two 8400 x 4725 files: 158 MB of pixels each; the per-format tests decode them)
        if O.stbi_load(b, 4, False) is None:
            continue
        files.append(b); exp.append(lambda rc, b=b: O.stbi_load(b, rc, False)[0].reshape(-1))
    for ch in (3, 4):
        img = rng.integers(0, 256, (53, 71, ch), dtype=np.uint8); img[10:30] = img[10:11, :1]
        q = gen.qoi_encode(img)
        files.append(q); exp.append(lambda rc, q=q: O.qoi_decode(q, rc)[0].reshape(-1))
    order = rng.permutation(len(files))
    return [files[i] for i in order], [exp[i] for i in order]


def _mixed_call(L, files, rc, stream=None):
    n = len(files)
    bufs = [np.frombuffer(f, np.uint8) if len(f) else np.zeros(1, np.uint8) for f in files]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); lens = (C.c_size_t * n)(*[len(f) for f in files])
    cap = 1 << 22                                               # a slot per file, large enough for every fixture
    offs = (np.arange(n, dtype=np.int64) * cap)
    dout = L.gamut_hip_device_malloc(n * cap)
    info = (_capi.ImageInfo * n)(); st = (C.c_int * n)()
    r = L.gamut_hip_decode_batch_device(ptrs, lens, n, rc, offs.ctypes.data_as(P64), dout, info, st, stream)
    host = _down(L, dout, n * cap)
    L.gamut_hip_device_free(dout)
    return r, list(st), info, [host[i * cap:(i + 1) * cap] for i in range(n)]


def test_mixed_format_batch_equals_oracle(hip):
    """gamut_hip_decode_batch_device: files of all three formats in ONE call, any order -- formats sniffed as identifyFormatFromStream does
    (image.d:1045-1061), the three pipelines side by side -- every file's pixels == the oracle's decoder for its format, for rgba8 and
    rgb8; an unknown file and a damaged one are reported per file and do not disturb the others; twice in a row (the worker threads and
    their staging buffers persist)"""
    files, exp = _mixed_files()
    files.insert(4, b"GIF89a not one of the three"); exp.insert(4, None)
    files.insert(9, files[0][:60]); exp.insert(9, None)
    for rc in (4, 3, 4):
        r, st, info, out = _mixed_call(hip, files, rc)
        assert r != 0 and st[4] == _capi.ERR_UNSUPPORTED and st[9] != 0 and info[4].format == -1
        for i, e in enumerate(exp):
            if e is None:
                continue
            want = e(rc)
            assert st[i] == 0, (i, st[i])
            assert info[i].width * info[i].height * rc == want.size and info[i].channels == rc, (i, info[i].width, info[i].height)
            assert np.array_equal(out[i][:want.size], want), (i, info[i].format)


def test_mixed_format_batch_from_two_threads(hip):
    """two caller threads at once: one gets the workers, the other runs its legs itself -- same pixels"""
    files, exp = _mixed_files()
    res = [None, None]

    def work(k):
        hip.gamut_hip_init(0)
        res[k] = _mixed_call(hip, files, 4)
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    for k in range(2):
        r, st, info, out = res[k]
        assert r == 0 and not any(st)
        for i, e in enumerate(exp):
            want = e(4)
            assert np.array_equal(out[i][:want.size], want), (k, i)
