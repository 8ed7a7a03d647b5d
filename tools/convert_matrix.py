#!/usr/bin/env python3
"""Throughput of every PixelType -> PixelType pair (k_convert_vec<S,D>) on one layer set, to spot outliers.
Prints the pairs sorted by algorithmic GB/s.  Usage: python tools/convert_matrix.py [side=4096] [layers=4]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from gamut_amd import _capi  # noqa: E402


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    npx = side * side
    src = torch.randint(0, 256, (layers * npx * 16,), device=dev, dtype=torch.int32).to(torch.uint8)      # any bytes are valid pixels except f32 NaNs: fine for timing
    dst = torch.empty((layers * npx * 16,), dtype=torch.uint8, device=dev)
    res = []
    for s, sn in enumerate(O.PIXEL_TYPES):
        for d, dn in enumerate(O.PIXEL_TYPES):
            if s == d:
                continue
            sp, dp = side * O.PT_SIZE[s], side * O.PT_SIZE[d]

            def step():
                _capi.check(L.gamut_hip_scanlines_convert_device(s, src.data_ptr(), sp, sp * side, d, dst.data_ptr(), dp, dp * side, side, side, layers, stream))
            step(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                step()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            res.append((layers * npx * (O.PT_SIZE[s] + O.PT_SIZE[d]) / ms / 1e6, sn, dn, ms))
    res.sort()
    for gbs, sn, dn, ms in res[:40]:
        print(f"{sn:>9s} -> {dn:<9s} {gbs:8.0f} GB/s  {ms:7.3f} ms")
    print("...")
    for gbs, sn, dn, ms in res[-5:]:
        print(f"{sn:>9s} -> {dn:<9s} {gbs:8.0f} GB/s  {ms:7.3f} ms")
    import statistics
    print(f"median {statistics.median(r[0] for r in res):.0f} GB/s over {len(res)} pairs")


if __name__ == "__main__":
    main()
