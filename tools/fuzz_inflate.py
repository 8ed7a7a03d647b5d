#!/usr/bin/env python3
"""Mutation fuzzing of the inflate kernel against zlib: damaged, truncated, spliced and random streams in batches; the device
must give zlib's verdict (and zlib's bytes when zlib accepts) and come back.  Run on the GPU box under `timeout`.
Usage: python tools/fuzz_inflate.py [iterations=4000] [seed=1] [slice_bytes=0: the sliced entry point, that many bytes of input per launch]
[seed_bytes=40000: uncompressed size of the seed streams -- 600000 makes streams of several rounds and tiles]"""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gamut_amd import _capi  # noqa: E402
from test_inflate_gpu import _corpus, _deflate, _inflate_device  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    slice_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    seed_bytes = int(sys.argv[4]) if len(sys.argv) > 4 else 40000
    corpus = {k: (v * (seed_bytes // len(v) + 1))[:seed_bytes] for k, v in _corpus().items() if len(v) > 100}
    seeds = []
    for name, data in corpus.items():
        for kw in (dict(level=6), dict(level=1), dict(level=9, mem=1), dict(level=6, strategy=zlib.Z_FIXED), dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY),
                   dict(level=0), dict(level=6, flush_every=500)):
            seeds.append(_deflate(data, **kw))
    cap = seed_bytes + 8000
    done = bad = accepted = 0
    t0 = time.time()
    while done < iters:
        cases = []
        for _ in range(256):
            s = bytearray(seeds[int(rng.integers(0, len(seeds)))])
            kind = int(rng.integers(0, 6))
            if kind == 0:                                            # bit flips
                for _ in range(int(rng.integers(1, 4))):
                    s[int(rng.integers(0, len(s)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:                                          # truncation
                s = s[:int(rng.integers(0, len(s)))]
            elif kind == 2:                                          # a run of random bytes
                i = int(rng.integers(0, len(s))); n = int(rng.integers(1, 64)); s[i:i + n] = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            elif kind == 3:                                          # splice of two streams
                o = seeds[int(rng.integers(0, len(seeds)))]; i = int(rng.integers(0, len(s))); j = int(rng.integers(0, len(o))); s = s[:i] + o[j:]
            elif kind == 4:                                          # pure noise
                s = bytearray(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8).tobytes())
            else:                                                    # damage near the start (block headers, code lengths)
                for _ in range(int(rng.integers(1, 3))):
                    s[int(rng.integers(0, min(len(s), 120)))] ^= 1 << int(rng.integers(0, 8))
            cases.append(bytes(s) if len(s) else b"\x00")
        rc, outs, st = _inflate_device(L, cases, [cap] * len(cases), slice_bytes)
        assert rc == 0
        for c, o, v in zip(cases, outs, st):
            d = zlib.decompressobj(-15)
            try:
                exp = d.decompress(c); ok = d.eof                    # the whole stream, whatever the capacity
            except zlib.error:
                ok = False; exp = b""
            if ok:
                accepted += 1
                if v != 0 or o != exp[:cap]:
                    bad += 1; print(f"MISMATCH: zlib accepts ({len(exp)} bytes), device status {v}, {len(o)} bytes; stream {c[:24].hex()}... len {len(c)}")
            elif v == 0:
                bad += 1; print(f"MISMATCH: zlib rejects, device accepted {len(o)} bytes; stream {c[:24].hex()}... len {len(c)}")
        done += len(cases)
    print(f"{done} streams ({accepted} valid for zlib), {bad} mismatches, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
