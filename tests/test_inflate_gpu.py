"""GPU parity: inflate on the device (gamut_hip_inflate_batch_device, inflate.hip) == zlib, byte for byte, on streams from every
encoder setting (levels, strategies, stored / fixed / dynamic blocks, many small blocks, window-sized distances, long runs), and
the same accept / reject decisions as zlib on damaged streams.  The PNG file batch then runs with the device inflate."""
import ctypes as C
import zlib

import numpy as np
import pytest

from gamut_amd import _capi

pytestmark = pytest.mark.gpu


def _deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, mem=8, flush_every=0):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    if not flush_every:
        return co.compress(data) + co.flush()
    out = []
    for i in range(0, len(data), flush_every):
        out.append(co.compress(data[i:i + flush_every]))
        out.append(co.flush(zlib.Z_FULL_FLUSH if (i // flush_every) % 2 else zlib.Z_SYNC_FLUSH))
    out.append(co.flush())
    return b"".join(out)


def _inflate_device(L, streams, caps, slice_bytes=0):
    """streams: list of raw-deflate byte strings; caps: output capacities -> (rc, [bytes], [status])"""
    n = len(streams)
    offs = np.concatenate([[0], np.cumsum([len(s) + 3 for s in streams])]).astype(np.int64)      # odd offsets: no alignment promised
    blob = np.zeros(int(offs[-1]) + 1, np.uint8)
    for i, s in enumerate(streams):
        blob[offs[i]:offs[i] + len(s)] = np.frombuffer(s, np.uint8)
    ooffs = np.concatenate([[0], np.cumsum([c + 5 for c in caps])]).astype(np.int64)
    dblob = L.gamut_hip_device_malloc(blob.size + 64); dout = L.gamut_hip_device_malloc(int(ooffs[-1]) + 64)
    dlen = L.gamut_hip_device_malloc(4 * n + 16); dst = L.gamut_hip_device_malloc(4 * n + 16)
    poison = np.full(int(ooffs[-1]) + 64, 0xA5, np.uint8)
    _capi.check(L.gamut_hip_memcpy_h2d(dblob, blob.ctypes.data, blob.size, None))
    _capi.check(L.gamut_hip_memcpy_h2d(dout, poison.ctypes.data, poison.size, None))
    descs = (_capi.InflateDesc * n)()
    for i in range(n):
        descs[i].src = dblob + int(offs[i]); descs[i].dst = dout + int(ooffs[i]); descs[i].src_len = len(streams[i]); descs[i].dst_cap = caps[i]
    _capi.check(L.gamut_hip_stream_synchronize(None))
    rc = L.gamut_hip_inflate_batch_device_sliced(descs, n, dlen, dst, slice_bytes, None) if slice_bytes else L.gamut_hip_inflate_batch_device(descs, n, dlen, dst, None)
    lens = np.zeros(n, np.uint32); st = np.zeros(n, np.uint32); host = np.empty(poison.size, np.uint8)
    _capi.check(L.gamut_hip_memcpy_d2h(lens.ctypes.data, dlen, 4 * n, None))
    _capi.check(L.gamut_hip_memcpy_d2h(st.ctypes.data, dst, 4 * n, None))
    _capi.check(L.gamut_hip_memcpy_d2h(host.ctypes.data, dout, host.size, None))
    _capi.check(L.gamut_hip_stream_synchronize(None))
    for p in (dblob, dout, dlen, dst):
        L.gamut_hip_device_free(p)
    outs = []
    for i in range(n):
        outs.append(host[ooffs[i]:ooffs[i] + lens[i]].tobytes())
        assert (host[ooffs[i] + caps[i]:ooffs[i + 1]] == 0xA5).all(), f"stream {i} wrote past its capacity"
    return rc, outs, list(st)


def _corpus():
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    text = (b"the quick brown fox jumps over the lazy dog. " * 3000)[:120000]
    grad = (np.add.outer(np.arange(300), np.arange(400)) // 3 % 256).astype(np.uint8).tobytes()
    lownoise = (rng.integers(0, 4, 200000, dtype=np.uint8) + (np.arange(200000) // 997 % 200).astype(np.uint8)).tobytes()
    sparse = np.where(rng.random(150000) < 0.02, rng.integers(1, 256, 150000), 0).astype(np.uint8).tobytes()
    far = noise[:32768] + noise[:32768] + noise[100:30000] + bytes(70000) + noise[:5000]         # distances up to the whole window
    return {"noise": noise, "text": text, "grad": grad, "lownoise": lownoise, "sparse": sparse, "far": far, "empty": b"", "one": b"x",
            "runs": b"a" * 100000 + b"ab" * 40000 + b"abc" * 30000}


def test_inflate_matches_zlib_on_every_encoder_setting(hip):
    corpus = _corpus()
    streams, expect, names = [], [], []
    for name, data in corpus.items():
        for level in (0, 1, 6, 9):
            streams.append(_deflate(data, level)); expect.append(data); names.append(f"{name} level {level}")
        for strat, sn in ((zlib.Z_FIXED, "fixed"), (zlib.Z_HUFFMAN_ONLY, "huffman-only"), (zlib.Z_RLE, "rle"), (zlib.Z_FILTERED, "filtered")):
            streams.append(_deflate(data, 6, strat)); expect.append(data); names.append(f"{name} {sn}")
        streams.append(_deflate(data, 6, flush_every=777)); expect.append(data); names.append(f"{name} flushed every 777")
        streams.append(_deflate(data, 9, mem=1)); expect.append(data); names.append(f"{name} memLevel 1 (small blocks)")
        streams.append(_deflate(data, 6, wbits=-9)); expect.append(data); names.append(f"{name} 512-byte window")
    rc, outs, st = _inflate_device(hip, streams, [len(e) + 7 for e in expect])
    assert rc == 0, hip.gamut_hip_last_error()
    for name, got, exp, s in zip(names, outs, expect, st):
        assert s == 0, f"{name}: status {s}"
        assert got == exp, f"{name}: {len(got)} bytes, expected {len(exp)}; first difference at {next((i for i, (a, b) in enumerate(zip(got, exp)) if a != b), None)}"


def test_inflate_capacity_clamps_and_large_stream(hip):
    rng = np.random.default_rng(5)
    big = (rng.integers(0, 16, 6_000_000, dtype=np.uint8) * 3 + (np.arange(6_000_000) // 4093 % 100).astype(np.uint8)).tobytes()
    s = _deflate(big, 6)
    # damage far beyond what a small capacity needs: still a corrupt stream (an invalid block type at the start of the last block
    # would be the surest damage; a flipped bit usually is one too -- zlib decides)
    damaged = bytearray(s); damaged[len(s) - 1000] ^= 0x40
    try:
        d = zlib.decompressobj(-15); d.decompress(bytes(damaged)); zlib_takes_it = d.eof
    except zlib.error:
        zlib_takes_it = False
    rc, outs, st = _inflate_device(hip, [s, s, s, bytes(damaged), s[:len(s) - 100]], [len(big), len(big) // 3 + 1, 0, 1000, 1000])
    assert rc == 0 and [int(v) for v in st[:3]] == [0, 0, 0], st
    assert (st[3] == 0) == zlib_takes_it and st[4] != 0, st
    assert outs[0] == big and outs[1] == big[:len(big) // 3 + 1] and outs[2] == b""


def test_inflate_rejects_what_zlib_rejects(hip):
    rng = np.random.default_rng(9)
    data = _corpus()["text"][:30000]
    good = _deflate(data, 6)
    cases = [good]
    for _ in range(60):                                                  # single-byte damage somewhere in the stream
        b = bytearray(good); i = int(rng.integers(0, len(b))); b[i] ^= 1 << int(rng.integers(0, 8)); cases.append(bytes(b))
    for cut in (1, 2, 5, len(good) // 2, len(good) - 1):                 # truncation
        cases.append(good[:cut])
    cases.append(b"\x07")                                                # block type 3
    cases.append(b"\x01\x05\x00\xfa\xfe" + b"abcde")                     # stored block with a bad ~LEN
    cases.append(b"\x01\x05\x00\xfa\xff" + b"abc")                       # stored block, data cut short
    rc, outs, st = _inflate_device(hip, cases, [len(data) + 64] * len(cases))
    assert rc == 0
    for i, c in enumerate(cases):
        d = zlib.decompressobj(-15)
        try:
            exp = d.decompress(c); ok = d.eof
        except zlib.error:
            ok = False
        if ok:
            assert st[i] == 0 and outs[i] == exp[:len(data) + 64], f"case {i}: zlib accepts, device status {st[i]}"
        else:
            assert st[i] != 0, f"case {i}: zlib rejects (or wants more input), the device accepted"


def test_inflate_many_small_blocks_and_the_block_budget(hip):
    """streams of many tiny blocks: every flavour of flush after every few bytes (accepted: far below the budget of a block per 8
    compressed bytes), a stream of 20 000 empty fixed-Huffman blocks (10 bits each: over the budget, turned away -- the bound on
    what a hostile stream can cost), and the same with an image's worth of data in front (still over)."""
    rng = np.random.default_rng(3)
    data = (rng.integers(0, 8, 30000, dtype=np.uint8) * 9).tobytes()
    flushed = _deflate(data, 6, flush_every=23)                      # ~1300 sync / full flushes: empty stored blocks
    fixed = _deflate(data, 6, zlib.Z_FIXED, flush_every=400)
    # 20 000 empty fixed blocks: BFINAL = 0, BTYPE = 01, end-of-block code 0000000 -> the bit string 0 10 0000000 repeated
    bits = np.tile(np.array([0, 1, 0, 0, 0, 0, 0, 0, 0, 0], np.uint8), 20000)
    last = np.array([1, 1, 0, 0, 0, 0, 0, 0, 0, 0], np.uint8)         # BFINAL = 1
    empty = np.packbits(np.concatenate([bits, last]), bitorder="little").tobytes()
    assert zlib.decompressobj(-15).decompress(empty) == b""
    rc, outs, st = _inflate_device(hip, [flushed, fixed, empty, empty[:-2]], [len(data) + 8, len(data) + 8, 64, 64])
    assert rc == 0
    assert st[0] == 0 and outs[0] == data and st[1] == 0 and outs[1] == data
    assert st[2] == 6 and st[3] == 6                                  # GAMUT_HIP_INFLATE_E_INPUT: over the block budget / cut short


def test_inflate_round_and_tile_boundaries(hip):
    """What the rounds / tiles of the kernel cut through (inflate.hip): runs far longer than a 28 KiB tile (their pieces copy from the tile's
    edge when the run began in front of it), one lane inflating to more than a tile (2-bit matches of 258 bytes), matches at the full
    32 KiB distance across tile and round boundaries, more tokens in a 32 KiB round than its list holds (1-bit literals: the round is
    cut at a lane), capacities that end inside a tile."""
    rng = np.random.default_rng(23)
    seed = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    two = rng.integers(0, 2, 3_000_000, dtype=np.uint8).tobytes()
    cases = {
        "zeros": bytes(5_000_000),
        "period 3": b"\x10\x80\xf0" * 1_200_000,
        "period 259": rng.integers(0, 256, 259, dtype=np.uint8).tobytes() * 12000,
        "period 32768": seed * 40,
        "period 32768 with edits": b"".join(seed[:k * 811 % 32768] + b"!" + seed[k * 811 % 32768 + 1:] for k in range(30)),
        "two symbols": two,
        "runs between noise": b"".join(rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() + bytes([int(n) & 255]) * int(40000 + n) for n in rng.integers(1, 3000, 25)),
    }
    streams, expect, caps, names = [], [], [], []
    for name, data in cases.items():
        for kw in (dict(level=6), dict(level=9), dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY), dict(level=6, strategy=zlib.Z_RLE), dict(level=1, mem=1)):
            if name == "two symbols" and kw.get("strategy") != zlib.Z_HUFFMAN_ONLY and kw.get("level") != 1:
                continue
            st = _deflate(data, **kw)
            for cap in (len(data), len(data) // 2 + 12345, 40000):
                streams.append(st); expect.append(data[:cap]); caps.append(cap); names.append(f"{name} {kw} cap {cap}")
    rc, outs, st = _inflate_device(hip, streams, caps)
    assert rc == 0, hip.gamut_hip_last_error()
    for name, got, exp, s_ in zip(names, outs, expect, st):
        assert s_ == 0, f"{name}: status {s_}"
        assert got == exp, f"{name}: {len(got)} bytes, expected {len(exp)}; first difference at {next((i for i, (a, b) in enumerate(zip(got, exp)) if a != b), None)}"


def test_inflate_in_slices_equals_inflate_at_once(hip):
    """gamut_hip_inflate_batch_device_sliced: the streams advance a slice of input per launch (what the PNG batch path does behind the
    upload) -- suspended in front of block headers, inside Huffman blocks (tables rebuilt from the saved code lengths, window from
    the output), in front of stored blocks; damaged streams keep their verdicts; capacities below the stream's size (the window is
    not reloaded once nothing is written any more)."""
    rng = np.random.default_rng(31)
    corpus = _corpus()
    big = (rng.integers(0, 16, 3_000_000, dtype=np.uint8) * 3 + (np.arange(3_000_000) // 4093 % 100).astype(np.uint8)).tobytes()
    noise = rng.integers(0, 256, 1_500_000, dtype=np.uint8).tobytes()
    streams, expect, caps = [], [], []
    for data in (big, noise, corpus["runs"] * 6, corpus["far"] * 5, bytes(2_000_000)):
        for kw in (dict(level=6), dict(level=1, mem=1), dict(level=0), dict(level=6, strategy=zlib.Z_FIXED), dict(level=6, flush_every=50000)):
            st = _deflate(data, **kw)
            for cap in (len(data), len(data) // 3):
                streams.append(st); expect.append(data[:cap]); caps.append(cap)
    good = _deflate(big, 6)
    bad = bytearray(good); bad[len(bad) // 2] ^= 0x55
    streams += [bytes(bad), good[:len(good) // 2]]; expect += [None, None]; caps += [len(big), len(big)]
    whole = _inflate_device(hip, streams, caps)
    for slice_bytes in (100_000, 40_000, 1_000_000):
        rc, outs, st = _inflate_device(hip, streams, caps, slice_bytes)
        assert rc == 0, hip.gamut_hip_last_error()
        for i, (got, exp, s_) in enumerate(zip(outs, expect, st)):
            if exp is None:
                assert (s_ != 0) == (whole[2][i] != 0), f"slice {slice_bytes}, stream {i}: status {s_}, at once {whole[2][i]}"
            else:
                assert s_ == 0 and got == exp, f"slice {slice_bytes}, stream {i}: status {s_}, {len(got)} of {len(exp)} bytes, first difference at {next((k for k, (a, b) in enumerate(zip(got, exp)) if a != b), None)}"


def test_scratch_follows_batches_of_growing_and_shrinking_size(hip):
    """the per-stream tables of a call lie behind the token lists the (thread, stream) scratch already owns: a batch of 17 streams sizes the lists
    for a full chip, a later batch of 150 must make the allocation grow with its tables (round 4 checked the size it needed against the token
    lists it would have needed, not the ones it had: the item table of the larger batch landed past the allocation)"""
    rng = np.random.default_rng(23)
    data = [rng.integers(0, 64, 3000 + 17 * i, dtype=np.uint8).tobytes() for i in range(150)]
    streams = [_deflate(d, 6) for d in data]
    for n in (17, 150, 5, 150, 40):
        rc, outs, st = _inflate_device(hip, streams[:n], [len(d) for d in data[:n]])
        assert rc == 0, hip.gamut_hip_last_error()
        assert not any(st)
        for i in range(n):
            assert outs[i] == data[i], (n, i)
