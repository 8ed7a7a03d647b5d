#!/usr/bin/env python3
"""Mutated QOI files through gamut_hip_qoi_decode_batch_device (every kernel: a batch of 24 takes the four-wave / pipelined kernels, a batch of 800
the one-wave kernel) against the oracle's qoi_decode (codecs/qoi.d:448-550): a file the oracle decodes must come out byte for byte (a
truncated stream decodes to the end of what is there and leaves the rest as the reference does), one it rejects must be rejected.
Usage: python tools/fuzz_qoi_gpu.py [batches=40] [seed=3]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
import oracle_lib as O  # noqa: E402
import torch  # noqa: E402,F401
from gamut_amd import _capi, synth  # noqa: E402


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    seeds = []
    for k, (w, h) in enumerate(((64, 48), (131, 17), (7, 200), (320, 96))):
        img = gen.synth_rgb(w, h, 40 + k)
        if k % 2: img[h // 3: h // 2] = img[h // 3, :1]                      # long runs
        seeds.append(bytes(synth.qoi_encode(img)))
        seeds.append(bytes(synth.qoi_encode(np.dstack([img, (img[:, :, 0] // 64 * 85)[:, :, None]]))))      # RGBA with a few alpha levels
    n_same = n_rej = 0
    for b in range(batches):
        n = 24 if b % 4 else 800
        ch = (4, 3, 0)[b % 3]
        blobs = []
        for k in range(n):
            data = bytearray(seeds[(b + k) % len(seeds)])
            if k % 5:
                for _ in range(int(rng.integers(1, 4))):
                    kind = int(rng.integers(0, 4))
                    if len(data) < 16: break
                    i = int(rng.integers(14 if kind < 3 else 0, len(data)))
                    if kind == 0: data[i] = int(rng.integers(0, 256))
                    elif kind == 1: data = data[:i]
                    elif kind == 2: data[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
                    else: data[i] = int(rng.integers(0, 256))
            blobs.append(bytes(data))
        exp = [O.qoi_decode(x, ch) if len(x) else None for x in blobs]
        sizes = [e[0].size if e is not None else 0 for e in exp]
        if sum(sizes) > (1 << 30):
            continue
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        out = torch.full((int(sum(sizes)) + 64,), 0xA5, dtype=torch.uint8, device="cuda")
        bufs = [np.frombuffer(x, np.uint8) if len(x) else np.zeros(1, np.uint8) for x in blobs]
        ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in bufs]); lens = (C.c_int * n)(*[len(x) for x in blobs])
        descs = (_capi.QoiDesc * n)(); hst = (C.c_int * n)()
        L.gamut_hip_qoi_decode_batch_device(ptrs, lens, n, ch, offs.ctypes.data_as(C.POINTER(C.c_int64)), out.data_ptr(), descs, hst, None)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i in range(n):
            if exp[i] is None:
                assert hst[i] != 0, f"batch {b} file {i}: the oracle rejects it, the library does not"
                n_rej += 1
                continue
            assert hst[i] == 0, f"batch {b} file {i}: rejected ({hst[i]}), the oracle decodes it"
            if not np.array_equal(got[offs[i]:offs[i] + sizes[i]], exp[i][0].reshape(-1)):
                open(os.path.join(ROOT, "gpurun_out", f"fuzz_qoi_mismatch_{b}_{i}.qoi"), "wb").write(blobs[i])
                bad = np.argwhere(got[offs[i]:offs[i] + sizes[i]] != exp[i][0].reshape(-1))
                raise AssertionError(f"batch {b} ({n} files, {ch} channels) file {i} ({len(blobs[i])} bytes, saved): {len(bad)} bytes differ, first at {int(bad[0][0])} of {sizes[i]}")
            n_same += 1
        assert (got[int(sum(sizes)):] == 0xA5).all(), "wrote past the last image"
    print(f"fuzz_qoi_gpu: {batches} batches: {n_same} files decoded like the oracle, {n_rej} rejected by both")


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    main()
