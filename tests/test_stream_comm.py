"""The callback-shaped drop-ins (stream_host.hip) and the C-ABI multi-GPU surface (comm.hip).

CPU part: shard arithmetic against gamut_amd/shard.py, argument validation, read-callback call pattern.
GPU part: *_from_stream / *_from_callbacks == the *_from_memory entry points == the oracle; world-1 gather; a 2-rank RCCL
gather on one box (both ranks on cuda:0 is refused by RCCL, so that test needs 2 devices and is skipped otherwise)."""
import ctypes as C
import os

import numpy as np
import pytest

import fixtures
import oracle_lib as O
from gamut_amd import _capi, shard

HERE = os.path.dirname(os.path.abspath(__file__))
JPEGS = fixtures.jpegs()
PNGS = fixtures.ref_pngs()


def _jpeg_reader(data, piece, raise_eof_late=False, fail_at=None):
    """A JpegStreamReadFunc over `data` handing out at most `piece` bytes per call (jpegload.d:61-70)."""
    state = {"pos": 0, "calls": 0}

    def rd(pbuf, max_bytes, peof, user):
        state["calls"] += 1
        if fail_at is not None and state["pos"] >= fail_at:
            return -1
        n = min(max_bytes, piece, len(data) - state["pos"])
        C.memmove(pbuf, data[state["pos"]:state["pos"] + n], n)
        state["pos"] += n
        if state["pos"] >= len(data) and not (raise_eof_late and n > 0):
            peof[0] = 1
        return n
    return _capi.JPEG_STREAM_READ_FUNC(rd), state


def _stb_callbacks(data, piece):
    state = {"pos": 0, "calls": 0}

    def rd(user, buf, size):
        state["calls"] += 1
        n = min(size, piece, len(data) - state["pos"])
        C.memmove(buf, data[state["pos"]:state["pos"] + n], n)
        state["pos"] += n
        return n

    def skip(user, n):
        state["pos"] += n
        state["skipped"] = state.get("skipped", 0) + n

    def eof(user):
        return int(state["pos"] >= len(data))
    cb = _capi.StbiIoCallbacks()
    keep = (type(cb.read)(rd), type(cb.skip)(skip), type(cb.eof)(eof))
    cb.read, cb.skip, cb.eof = keep
    return cb, state, keep


# ------------------------------------------------------------------------------------------------ CPU
def test_shard_arithmetic_matches_python_sharder():
    L = _capi.lib()
    for world in (1, 2, 3, 8):
        for total in (0, 1, 7, 8, 8191, 8192):
            seen = []
            for rank in range(world):
                idx = shard.shard_indices(total, rank, world)
                assert L.gamut_hip_shard_count(rank, world, total) == len(idx)
                for k, g in enumerate(idx):
                    assert L.gamut_hip_shard_owner(g, world) == rank
                    assert L.gamut_hip_shard_local_index(g, world) == k
                    assert L.gamut_hip_shard_global_index(k, rank, world) == g
                seen += list(idx)
            assert sorted(seen) == list(range(total))
    assert L.gamut_hip_shard_owner(-1, 4) == -1 and L.gamut_hip_shard_owner(3, 0) == -1
    assert L.gamut_hip_shard_count(4, 4, 10) == -1 and L.gamut_hip_shard_count(0, 0, 10) == -1


def test_comm_world1_and_argument_validation_need_no_rccl():
    L = _capi.lib()
    comm = C.c_void_p()
    assert L.gamut_hip_comm_init(C.byref(comm), 1, 0, None) == 0
    assert L.gamut_hip_comm_rank(comm) == 0 and L.gamut_hip_comm_world(comm) == 1
    # nothing to move: returns before touching the device
    assert L.gamut_hip_gather_outputs_device(comm, None, 0, 0, 0, None, 0, -1, None) == 0
    assert L.gamut_hip_gather_outputs_device(comm, None, 16, 16, 4, None, 16, 0, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_gather_outputs_device(comm, None, 8, 16, 4, None, 16, 0, None) == _capi.ERR_INVALID_ARG     # stride < image
    assert L.gamut_hip_gather_outputs_device(comm, None, 16, 16, 4, None, 16, 1, None) == _capi.ERR_INVALID_ARG    # root >= world
    L.gamut_hip_comm_destroy(comm)
    assert L.gamut_hip_comm_init(C.byref(comm), 2, 2, None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_comm_init(C.byref(comm), 2, 0, None) == _capi.ERR_INVALID_ARG                               # world 2 needs an id
    assert L.gamut_hip_comm_get_unique_id(None) == _capi.ERR_INVALID_ARG
    assert L.gamut_hip_comm_rank(None) == -1
    assert L.gamut_hip_host_threads() >= 1


def test_host_threads_divide_by_local_ranks(tmp_path):
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); from gamut_amd import _capi; print(_capi.lib().gamut_hip_host_threads())"
            % os.path.dirname(HERE))
    def run(env):
        e = dict(os.environ); e.pop("GAMUT_HIP_HOST_THREADS", None); e.pop("LOCAL_WORLD_SIZE", None); e.update(env)
        return int(subprocess.check_output([sys.executable, "-c", code], env=e).split()[-1])
    whole = run({})
    assert run({"LOCAL_WORLD_SIZE": "2"}) == (whole + 1) // 2
    assert run({"LOCAL_WORLD_SIZE": "1024"}) == 1
    assert run({"GAMUT_HIP_HOST_THREADS": "3", "LOCAL_WORLD_SIZE": "2"}) == 3


def test_stream_entry_points_reject_null_callbacks():
    L = _capi.lib()
    w, h, ac = C.c_int(), C.c_int(), C.c_int()
    assert not L.gamut_hip_decompress_jpeg_image_from_stream(None, None, C.byref(w), C.byref(h), C.byref(ac), None, None, 4)
    assert L.gamut_hip_last_error() != b""
    assert not L.gamut_hip_stbi_load_from_callbacks(None, None, C.byref(w), C.byref(h), C.byref(ac), 4, None, None, None)
    assert not L.gamut_hip_stbi_load_16_from_callbacks(None, None, C.byref(w), C.byref(h), C.byref(ac), 4, None, None, None)
    assert L.gamut_hip_stbi_png_is16_from_callbacks(None, None) == 0
    # a read function that reports an error: NULL + message (stop_decoding(JPGD_STREAM_READ), jpegload.d:1994)
    rd, _ = _jpeg_reader(b"\xff\xd8" + bytes(100), 16, fail_at=32)
    assert not L.gamut_hip_decompress_jpeg_image_from_stream(rd, None, C.byref(w), C.byref(h), C.byref(ac), None, None, 4)
    assert b"read error" in L.gamut_hip_last_error()


def test_png_is16_from_callbacks_reads_the_header_only():
    L = _capi.lib()
    for path in PNGS:
        data = open(path, "rb").read()
        buf = np.frombuffer(data, np.uint8)
        cb, st, keep = _stb_callbacks(data, 7)
        assert L.gamut_hip_stbi_png_is16_from_callbacks(C.byref(cb), None) == L.gamut_hip_png_is16(buf.ctypes.data, buf.size)
        assert st["pos"] <= 64, "is16 must not consume the stream beyond the header"


def _with_ancillary_chunk(png, payload_len=5000):
    """the file with a tEXt chunk behind IHDR: stb skips such chunks through the `skip` callback (stbdec.d:822-842, 2018)"""
    import struct, zlib
    body = b"tEXt" + bytes(range(256)) * (payload_len // 256) + bytes(payload_len % 256)
    chunk = struct.pack(">I", len(body) - 4) + body + struct.pack(">I", zlib.crc32(body))
    at = png.index(b"IHDR") + 4 + 13 + 4                       # behind IHDR's CRC (CgBI files have a chunk before IHDR)
    return png[:at] + chunk + png[at:]


def _png_end(data):
    """offset behind IEND's CRC, or len(data) for the files that have none (issue #92)"""
    at = data.find(b"IEND")
    return at + 8 if at >= 0 and at + 8 <= len(data) else len(data)


def test_png_callbacks_stop_behind_iend_and_skip_ancillary_chunks():
    """An image embedded in a longer stream (Image.loadFromStream, image.d:916): the walk reads what stbi__parse_png_file
    parses (stbdec.d:1777-2023) and leaves the stream at the end of the image.  Runs without a GPU: the gathering happens
    before any device work (the decode itself then fails with NO_DEVICE here; the GPU twin below checks the pixels)."""
    L = _capi.lib()
    x, y, n = C.c_int(), C.c_int(), C.c_int()
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    for path in PNGS:
        first = open(path, "rb").read()
        for data in (first, _with_ancillary_chunk(first)):
            end = _png_end(data)
            stream = data + open(PNGS[0], "rb").read() + bytes(70000)
            if end == len(data):
                stream = data                                    # no IEND: the image ends with the stream
            for piece in (1 << 20, 1000, 7):
                cb, st, keep = _stb_callbacks(stream, piece)
                p = L.gamut_hip_stbi_load_from_callbacks(C.byref(cb), None, C.byref(x), C.byref(y), C.byref(n), 4, None, None, None)
                if p:
                    libc.free(p)
                assert st["pos"] == end, (os.path.basename(path), piece, st["pos"], end)
                if data is not first:
                    assert st.get("skipped", 0) >= 5000, "ancillary chunks are skipped, not read"


def test_jpeg_stream_stops_at_eoi():
    """jpgd reads 8 KiB pieces and stops decoding at EOI (jpegload.d:1971-2003): the gatherer makes no call once EOI is in,
    so at most one piece is over-read -- the reference's own granularity.  Embedded FFD9 inside an APPn payload is not EOI."""
    L = _capi.lib()
    w, h, ac = C.c_int(), C.c_int(), C.c_int()
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    for path in JPEGS:
        data = open(path, "rb").read()
        end = data.rfind(b"\xff\xd9") + 2
        assert end >= 2
        thumb = data[:2] + b"\xff\xe1\x00\x08\xff\xd9\xff\xd9\x00\x00" + data[2:]           # an APP1 segment holding FFD9 FFD9
        for d, e in ((data, end), (thumb, end + 10)):
            stream = d[:e] + bytes(100000)
            for piece in (1 << 20, 977):
                rd, st = _jpeg_reader(stream, piece)
                p = L.gamut_hip_decompress_jpeg_image_from_stream(rd, None, C.byref(w), C.byref(h), C.byref(ac), None, None, 4)
                if p:
                    libc.free(p)
                assert e <= st["pos"] < e + 8192, (os.path.basename(path), piece, st["pos"], e)


def test_png_callbacks_reject_an_ancillary_chunk_in_front_of_ihdr():
    """stbi__parse_png_file's `first` (stbdec.d:2003-2006): only IHDR (CgBI in front of it) may come first.  The walk keeps the offending
    chunk header so that the memory parser rejects the file the way gamut_hip_stbi_load_from_memory does; an ancillary chunk with an
    over-long length keeps its header too (no clean end of data behind IDAT that the issue-#92 path could accept)."""
    import struct, zlib
    L = _capi.lib()
    png = open(fixtures.ref_image("issue65.png"), "rb").read()
    body = b"tEXtComment\0hello"
    chunk = struct.pack(">I", len(body) - 4) + body + struct.pack(">I", zlib.crc32(body))
    bad = png[:8] + chunk + png[8:]
    x, y, n = C.c_int(), C.c_int(), C.c_int()
    cb, st, keep = _stb_callbacks(bad, 1 << 20)
    assert not L.gamut_hip_stbi_load_from_callbacks(C.byref(cb), None, C.byref(x), C.byref(y), C.byref(n), 4, None, None, None)
    assert b"first not IHDR" in L.gamut_hip_last_error()
    buf = np.frombuffer(bad, np.uint8)
    assert not L.gamut_hip_stbi_load_from_memory(buf.ctypes.data, buf.size, C.byref(x), C.byref(y), C.byref(n), 4, None, None, None)
    assert b"first not IHDR" in L.gamut_hip_last_error()
    assert st["pos"] == 16, "the walk stops at the offending chunk header"
    # an over-long ancillary chunk behind the last IDAT: its header must arrive at the memory parser
    at = png.index(b"IEND") - 4
    huge = png[:at] + struct.pack(">I", 0x80000001) + b"tEXt" + png[at:]
    cb, st, keep = _stb_callbacks(huge, 1 << 20)
    p = L.gamut_hip_stbi_load_from_callbacks(C.byref(cb), None, C.byref(x), C.byref(y), C.byref(n), 4, None, None, None)
    assert st["pos"] == at + 8, "the walk stops behind the over-long header and keeps it"
    if p:
        C.CDLL(None).free(C.c_void_p(p))


def _build_callback_consumer(tmp_path):
    import subprocess
    exe = str(tmp_path / "callback_consumer")
    root = os.path.dirname(HERE)
    lib_dir = os.path.dirname(_capi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(HERE, "c", "callback_consumer.c"), "-o", exe, "-L", lib_dir, "-lgamut_hip", "-Wl,-rpath," + lib_dir])
    return exe


def _consumer_stream(tmp_path):
    a = open(fixtures.ref_image("issue65.png"), "rb").read()
    b = _with_ancillary_chunk(open(fixtures.ref_image("issue76.png"), "rb").read())        # 16-bit grey: the _16_ entry point
    j = open(JPEGS[0], "rb").read()
    path = str(tmp_path / "three_images.bin")
    open(path, "wb").write(a + b + j)
    return path, (a, b, j)


def test_c_callbacks_with_the_reference_struct_layouts_walk_the_stream(tmp_path):
    """tests/c/callback_consumer.c: IOStream / IOAndHandle / JPEGIOHandle / stbi_io_callbacks with the reference's layouts, the callback
    bodies of stbdec.d:143-165 and plugins/jpeg.d:167-177 with C linkage (what bindings/gamut_hip.d's trampolines are), stdio underneath.
    Without a GPU the decode itself reports NO_DEVICE, but the stream has been walked by then: positions and call kinds are checked."""
    import subprocess
    exe = _build_callback_consumer(tmp_path)
    path, (a, b, j) = _consumer_stream(tmp_path)
    out = subprocess.run([exe, path, "4"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    lines = [l.split() for l in out.stdout.strip().splitlines()]
    assert [l[0] for l in lines] == ["png", "png", "jpeg"]
    assert int(lines[0][7]) == len(a) and int(lines[1][7]) == len(a) + len(b)             # left right behind IEND's CRC
    assert int(lines[1][4]) == 1 and int(lines[1][9]) >= 1                                  # is16; the tEXt chunk went through `skip`
    assert len(a) + len(b) + len(j) == int(lines[2][7])                                     # the JPEG is the end of the file


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_c_callbacks_with_the_reference_struct_layouts_decode(hip, tmp_path):
    """the same program on the GPU box: pixels == the oracle (FNV-1a of the decoded bytes), for req_comp 4 and 0 / -1"""
    import subprocess
    exe = _build_callback_consumer(tmp_path)
    path, imgs = _consumer_stream(tmp_path)

    def fnv(b):
        h = 1469598103934665603
        for v in np.frombuffer(b, np.uint8).tolist():
            h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h
    for req in (4, 0):
        out = subprocess.run([exe, path, str(req)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        lines = [l.split() for l in out.stdout.strip().splitlines()]
        assert len(lines) == 3
        for l, data in zip(lines, imgs):
            if l[0] == "png":
                exp, comp = O.stbi_load(data, req, l[4] == "1")[:2]
            else:
                e = O.decompress_jpeg(data, req if req else -1)
                exp, comp = e[0], e[1]
            assert (int(l[2]), int(l[3])) == (exp.shape[0], comp), l
            assert int(l[5]) == exp.nbytes and int(l[6], 16) == fnv(np.ascontiguousarray(exp).tobytes()), l


@pytest.mark.gpu
def test_jpeg_from_stream_on_the_input_layer_files(hip):
    """decompress_jpeg_image_from_stream's drop-in on tests/golden/jpeg_fuzz, in pieces of 8192 and of 501 bytes: NULL / pixels / density as
    expected.json has them (the stream walk must not stop early on bytes in front of SOI, nor at an EOI inside a segment)"""
    import json
    d = os.path.join(HERE, "golden", "jpeg_fuzz")
    expected = json.load(open(os.path.join(d, "expected.json")))
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    for name, e in sorted(expected.items()):
        data = open(os.path.join(d, name), "rb").read()
        for piece in (8192, 501):
            rd, st = _jpeg_reader(data, piece)
            w, h, ac = C.c_int(), C.c_int(), C.c_int()
            par, dpi = C.c_float(), C.c_float()
            p = hip.gamut_hip_decompress_jpeg_image_from_stream(rd, None, C.byref(w), C.byref(h), C.byref(ac), C.byref(par), C.byref(dpi), 3)
            assert bool(p) == (e["verdict"] == "image"), (name, piece, hip.gamut_hip_last_error())
            if not p:
                continue
            got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value * 3)).copy()
            libc.free(p)
            assert np.array_equal(got, O.decompress_jpeg(data, 3)[0]), (name, piece)
            exp = tuple(float("nan") if v == "nan" else v for v in (e["pixel_aspect_ratio"], e["dpi_y"]))
            assert O.same_density((par.value, dpi.value), exp), (name, piece)


@pytest.mark.gpu
@pytest.mark.parametrize("path", JPEGS, ids=[os.path.basename(p) for p in JPEGS])
def test_jpeg_from_stream_equals_oracle(hip, path):
    data = open(path, "rb").read()
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    for piece, late in ((1 << 20, False), (4096, False), (977, True)):
        for rc in (-1, 4):
            exp = O.decompress_jpeg(data, rc)
            rd, st = _jpeg_reader(data, piece, raise_eof_late=late)
            w, h, ac = C.c_int(), C.c_int(), C.c_int()
            par, dpi = C.c_float(), C.c_float()
            p = hip.gamut_hip_decompress_jpeg_image_from_stream(rd, None, C.byref(w), C.byref(h), C.byref(ac), C.byref(par), C.byref(dpi), rc)
            assert p, hip.gamut_hip_last_error()
            comps = ac.value if rc < 0 else rc
            got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value * comps)).copy()
            libc.free(p)
            assert np.array_equal(got, exp[0])
            assert ac.value == exp[1] and O.same_density((par.value, dpi.value), exp[2:])
            assert st["pos"] == len(data)                      # the fixtures end with EOI: nothing behind the image
    # optional out-pointers may be NULL, as in the reference's callers that do not want the DPI
    rd, _ = _jpeg_reader(data, 1 << 16)
    p = hip.gamut_hip_decompress_jpeg_image_from_stream(rd, None, C.byref(w), C.byref(h), C.byref(ac), None, None, 4)
    assert p
    libc.free(p)


@pytest.mark.gpu
@pytest.mark.parametrize("path", PNGS, ids=[os.path.basename(p) for p in PNGS])
def test_png_from_callbacks_equals_oracle(hip, path):
    data = open(path, "rb").read()
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    for sixteen in (False, True):
        for req in (0, 4):
            exp = O.stbi_load(data, req, sixteen)
            cb, st, keep = _stb_callbacks(data, 1000)
            x, y, n = C.c_int(), C.c_int(), C.c_int()
            fx, fy, fr = C.c_float(), C.c_float(), C.c_float()
            fn = hip.gamut_hip_stbi_load_16_from_callbacks if sixteen else hip.gamut_hip_stbi_load_from_callbacks
            p = fn(C.byref(cb), None, C.byref(x), C.byref(y), C.byref(n), req, C.byref(fx), C.byref(fy), C.byref(fr))
            if exp is None:
                assert not p
                continue
            assert p, hip.gamut_hip_last_error()
            comps = n.value if req == 0 else req
            ct = C.c_uint16 if sixteen else C.c_uint8
            got = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (y.value, x.value, comps)).copy()
            libc.free(p)
            assert n.value == exp[1] and np.array_equal(got, exp[0])


@pytest.mark.gpu
def test_two_images_back_to_back_in_one_stream(hip):
    """PNG, PNG (with a skipped chunk), JPEG one behind the other in one stream: each call decodes its image == the oracle and
    leaves the stream where the next one starts (stbdec.d:754-770,1777-2023; the JPEG comes last: its reader over-reads < 8 KiB)."""
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    a = open(fixtures.ref_image("issue65.png"), "rb").read()
    b = _with_ancillary_chunk(open(fixtures.ref_image("vst3-compatible.png"), "rb").read())
    j = open(JPEGS[0], "rb").read()
    stream = a + b + j + bytes(50000)
    cb, st, keep = _stb_callbacks(stream, 1000)
    for png, start in ((a, 0), (b, len(a))):
        assert st["pos"] == start
        exp = O.stbi_load(png, 4, False)
        x, y, n = C.c_int(), C.c_int(), C.c_int()
        p = hip.gamut_hip_stbi_load_from_callbacks(C.byref(cb), None, C.byref(x), C.byref(y), C.byref(n), 4, None, None, None)
        assert p, hip.gamut_hip_last_error()
        got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (y.value, x.value, 4)).copy()
        libc.free(p)
        assert n.value == exp[1] and np.array_equal(got, exp[0])
    assert st["pos"] == len(a) + len(b)

    def rd(pbuf, max_bytes, peof, user):                         # the JPEG reader continues on the same stream state
        k = min(max_bytes, len(stream) - st["pos"])
        C.memmove(pbuf, stream[st["pos"]:st["pos"] + k], k)
        st["pos"] += k
        if st["pos"] >= len(stream):
            peof[0] = 1
        return k
    w, h, ac = C.c_int(), C.c_int(), C.c_int()
    p = hip.gamut_hip_decompress_jpeg_image_from_stream(_capi.JPEG_STREAM_READ_FUNC(rd), None, C.byref(w), C.byref(h), C.byref(ac), None, None, 4)
    assert p, hip.gamut_hip_last_error()
    exp = O.decompress_jpeg(j, 4)
    got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value * 4)).copy()
    libc.free(p)
    assert np.array_equal(got, exp[0])
    assert len(a) + len(b) + len(j) <= st["pos"] < len(a) + len(b) + len(j) + 8192


@pytest.mark.gpu
def test_gather_world1_places_images_in_batch_order(hip):
    L = hip
    comm = C.c_void_p()
    assert L.gamut_hip_comm_init(C.byref(comm), 1, 0, None) == 0
    n, b, ls, ds = 5, 1000, 1024, 1500
    src = np.random.default_rng(3).integers(0, 256, n * ls, dtype=np.uint8)
    dsrc = C.c_void_p(L.gamut_hip_device_malloc(src.nbytes)); ddst = C.c_void_p(L.gamut_hip_device_malloc(n * ds))
    _capi.check(L.gamut_hip_memcpy_h2d(dsrc, src.ctypes.data, src.nbytes, None))
    _capi.check(L.gamut_hip_gather_outputs_device(comm, dsrc, ls, b, n, ddst, ds, -1, None))
    out = np.zeros(n * ds, np.uint8)
    _capi.check(L.gamut_hip_memcpy_d2h(out.ctypes.data, ddst, out.nbytes, None))
    _capi.check(L.gamut_hip_stream_synchronize(None))
    for i in range(n):
        assert np.array_equal(out[i * ds:i * ds + b], src[i * ls:i * ls + b])
    L.gamut_hip_device_free(dsrc); L.gamut_hip_device_free(ddst); L.gamut_hip_comm_destroy(comm)


def _build_shard_host(tmp_path):
    import subprocess
    exe = str(tmp_path / "shard_host")
    root = os.path.dirname(HERE)
    lib_dir = os.path.dirname(_capi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), os.path.join(HERE, "c", "shard_host.c"),
                           "-o", exe, "-L", lib_dir, "-lgamut_hip", "-lpthread", "-Wl,-rpath," + lib_dir])
    return exe


def test_pure_c_shard_host_builds(tmp_path):
    """tests/c/shard_host.c (shard -> decode its share -> gather, no Python in the ranks) compiles against the C ABI; without a GPU it
    reports that there is none"""
    import subprocess
    exe = _build_shard_host(tmp_path)
    if _capi.lib().gamut_hip_device_count() == 0:
        out = subprocess.run([exe, "proc", "1", "0", "-"], capture_output=True, text=True, timeout=60)
        assert out.returncode == 9 and "no device" in out.stderr


@pytest.mark.gpu
def test_pure_c_shard_host(hip, tmp_path):
    """The multi-GPU path driven by a C host, as a D host would: world 1 on any box; on a box with >= 2 devices ALSO as one
    process per GPU (RCCL id through a file) and as one process with one host thread per GPU -- this test does not skip there,
    so the first multi-GPU node the suite meets exercises RCCL."""
    import subprocess
    exe = _build_shard_host(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for mode in (["proc", "1", "0", "-"], ["threads", "1"]):
        out = subprocess.run([exe] + mode, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0 and "ok" in out.stdout, (mode, out.returncode, out.stdout, out.stderr)
    ndev = hip.gamut_hip_device_count()
    if ndev >= 2:
        world = min(ndev, 4)
        idfile = str(tmp_path / "rccl_id")
        ps = [subprocess.Popen([exe, "proc", str(world), str(r), idfile], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
        res = [p.communicate(timeout=600) + (p.returncode,) for p in ps]
        assert all(rc == 0 for _, _, rc in res), res
        out = subprocess.run([exe, "threads", str(world)], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0 and f"{world} threads ok" in out.stdout, (out.returncode, out.stdout, out.stderr)


def _build_rccl_double(tmp_path):
    """tests/c/rccl_double.c: the eight nccl* entry points comm.hip binds, for ranks that are threads of one process on one device"""
    import subprocess
    so = str(tmp_path / "librccl_double.so")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(HERE, "c", "rccl_double.c"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"])
    return so


def test_rccl_library_override(tmp_path):
    """GAMUT_HIP_RCCL_LIB names the library comm.hip binds (comm.hip: rccl()): the test double exports every symbol and hands out an id
    without a device; a path that does not exist is an error that says so -- never a silent fall-back to another copy of RCCL."""
    import subprocess
    import sys
    so = _build_rccl_double(tmp_path)
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert f" T {name}" in syms, name
    prog = ("import ctypes as C, sys; sys.path.insert(0, %r); from gamut_amd import _capi; L = _capi.lib(); b = (C.c_uint8 * 128)(); "
            "rc = L.gamut_hip_comm_get_unique_id(b); print(rc, bytes(b[:11]), _capi.last_error())" % os.path.dirname(HERE))
    out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120, env=dict(os.environ, GAMUT_HIP_RCCL_LIB=so))
    assert out.returncode == 0 and out.stdout.startswith("0 b'rccl-double'"), (out.stdout, out.stderr)
    out = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120, env=dict(os.environ, GAMUT_HIP_RCCL_LIB=str(tmp_path / "no_such.so")))
    assert out.returncode == 0 and not out.stdout.startswith("0 ") and "could not be loaded" in out.stdout and "no_such.so" in out.stdout, (out.stdout, out.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_gather_many_ranks_on_one_device(hip, tmp_path, world):
    """gamut_hip_gather_outputs_device with world > 1 on a 1-GPU box: tests/c/shard_host.c's threads form -- a host
    thread per rank, every rank on device 0 -- over tests/c/rccl_double.c instead of librccl (real RCCL refuses two ranks on one
    device).  Everything above the transport is the product's: which image goes to whom (round-robin owners, root / every rank), the
    own images' strided device copy, grouped ncclSend / ncclRecv in groups of 256 (600 and 257 images cross the group size), strides
    larger than an image.  shard_host checks every byte of every image on every rank that receives, that the gaps between images
    keep their fill, and that a rank that receives nothing is not written at all; roots: every rank, rank 0, the last rank."""
    import subprocess
    exe = _build_shard_host(tmp_path)
    so = _build_rccl_double(tmp_path)
    env = dict(os.environ, GAMUT_HIP_RCCL_LIB=so, HIP_VISIBLE_DEVICES="0")
    for total, pad in ((0, 0), (1, 48), (7, 0), (8, 48), (257, 0), (600, 48)):
        out = subprocess.run([exe, "threads", str(world), str(total), str(pad)], capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0 and f"{world} threads ok ({total} images)" in out.stdout, (world, total, pad, out.returncode, out.stdout, out.stderr[-2000:])


def _rccl_rank(rank, world, idfile, ndev, q):
    """One rank of the C-ABI gather: no torch.distributed -- the id travels through a file, as a D host would do it."""
    import time
    try:
        L = _capi.lib()
        _capi.check(L.gamut_hip_init(rank % ndev))
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _capi.check(L.gamut_hip_comm_get_unique_id(ident))
            with open(idfile + ".tmp", "wb") as f:
                f.write(bytes(ident))
            os.rename(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile):
                time.sleep(0.05)
                assert time.time() - t0 < 120
            C.memmove(ident, open(idfile, "rb").read(), 128)
        comm = C.c_void_p()
        _capi.check(L.gamut_hip_comm_init(C.byref(comm), world, rank, ident))
        total, b = 11, 4096 + 13
        mine = L.gamut_hip_shard_count(rank, world, total)
        rng = [np.random.default_rng(100 + i).integers(0, 256, b, dtype=np.uint8) for i in range(total)]
        local = np.concatenate([rng[L.gamut_hip_shard_global_index(k, rank, world)] for k in range(mine)])
        dloc = C.c_void_p(L.gamut_hip_device_malloc(local.nbytes)); ddst = C.c_void_p(L.gamut_hip_device_malloc(total * b))
        _capi.check(L.gamut_hip_memcpy_h2d(dloc, local.ctypes.data, local.nbytes, None))
        for root in (-1, 1):
            zero = np.zeros(total * b, np.uint8)
            _capi.check(L.gamut_hip_memcpy_h2d(ddst, zero.ctypes.data, zero.nbytes, None))
            _capi.check(L.gamut_hip_gather_outputs_device(comm, dloc, b, b, total, ddst, b, root, None))
            _capi.check(L.gamut_hip_stream_synchronize(None))
            if root < 0 or root == rank:
                out = np.zeros(total * b, np.uint8)
                _capi.check(L.gamut_hip_memcpy_d2h(out.ctypes.data, ddst, out.nbytes, None))
                _capi.check(L.gamut_hip_stream_synchronize(None))
                assert np.array_equal(out, np.concatenate(rng)), f"rank {rank} root {root}"
        L.gamut_hip_comm_destroy(comm)
        q.put((rank, "ok"))
    except BaseException as e:                                  # noqa: BLE001 -- reported to the parent
        q.put((rank, repr(e)))


@pytest.mark.gpu
def test_rccl_gather_two_ranks_through_the_c_abi(hip, tmp_path):
    import multiprocessing as mp
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("RCCL refuses two ranks on one device; needs 2 GPUs (the driver's 8-GPU node)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = str(tmp_path / "rccl_id")
    ps = [ctx.Process(target=_rccl_rank, args=(r, 2, idfile, ndev, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=300) for _ in ps)
    [p.join(60) for p in ps]
    assert res == {0: "ok", 1: "ok"}, res
