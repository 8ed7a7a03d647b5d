#!/usr/bin/env python3
"""Throughput of every PixelType -> PixelType pair (k_convert_vec<S,D>) on one layer set, to spot outliers.
Prints the pairs sorted by algorithmic GB/s.  Usage: python tools/convert_matrix.py [side=4096] [layers=4]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gamut_amd import _capi  # noqa: E402


PIXEL_TYPES = ['l8', 'l16', 'lf32', 'la8', 'la16', 'laf32', 'lap8', 'lap16', 'lapf32', 'rgb8', 'rgb16', 'rgbf32', 'rgba8', 'rgba16', 'rgbaf32',
               'rgbap8', 'rgbap16', 'rgbapf32']                    # PixelType ordinals, types.d:32-59
PT_SIZE = [1, 2, 4, 2, 4, 8, 2, 4, 8, 3, 6, 12, 4, 8, 16, 4, 8, 16]


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
    for s, sn in enumerate(PIXEL_TYPES):
        for d, dn in enumerate(PIXEL_TYPES):
            if s == d:
                continue
            sp, dp = side * PT_SIZE[s], side * PT_SIZE[d]

            def step():
                _capi.check(L.gamut_hip_scanlines_convert_device(s, src.data_ptr(), sp, sp * side, d, dst.data_ptr(), dp, dp * side, side, side, layers, stream))
            step(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                step()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            res.append((layers * npx * (PT_SIZE[s] + PT_SIZE[d]) / ms / 1e6, sn, dn, ms))
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
