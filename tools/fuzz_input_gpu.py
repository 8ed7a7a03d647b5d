#!/usr/bin/env python3
"""fuzz_input_gpu.py -- the corpus and the mutations of tools/fuzz_input.py (markers, tables, EXIF, restart structures, truncation, bytes in
front of SOI and behind the scan) through the DEVICE decoders: batches of mutated baseline / progressive files through
gamut_hip_jpeg_entropy_decode_device (verdict, every coefficient, every max_zag, pixelAspectRatio / dotsPerInchY == oracle) and
gamut_hip_jpeg_decode_batch_device (verdict and pixels == oracle), with the progressive scans decoded on the device and on the host, the
scans unstuffed on the device and on the host.  What a kernel flags is decoded again by the host feeder (host_redo): the run also reports how
many files went that way.
    python tools/fuzz_input_gpu.py [batches=40] [seed=1]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import fuzz_input as F            # noqa: E402
import oracle_lib as O            # noqa: E402
import torch                      # noqa: E402,F401
from gamut_amd import _capi       # noqa: E402
import test_jpeg_gpu as T         # noqa: E402  (its call helpers)


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    L = _capi.lib(); _capi.check(L.gamut_hip_init(0))
    pool = F.seeds(0)
    rng = np.random.default_rng([seed, 77])
    n_img = n_null = n_files = 0
    for b in range(batches):
        os.environ["GAMUT_HIP_JPEG_PROGRESSIVE"] = ("device", "host")[b & 1]
        os.environ["GAMUT_HIP_JPEG_UNSTUFF"] = ("device", "host", "")[b % 3]
        blobs = []
        while len(blobs) < 64:
            name, data = pool[int(rng.integers(0, len(pool)))]
            m = data if len(blobs) % 8 == 0 else F.mutate(data, rng)
            fr = _capi.JpegFrame(); buf = np.frombuffer(m, np.uint8) if m else np.zeros(1, np.uint8)
            if L.gamut_hip_jpeg_read_header(buf.ctypes.data, len(m), C.byref(fr)) == 0 and fr.width * fr.height > 1 << 20:
                continue                                           # a damaged frame header asking for megapixels: not this tool's subject
            blobs.append(m)
        expect = []
        for m in blobs:
            try:
                expect.append(O.DecodedJpeg(m))
            except ValueError:
                expect.append(None)
        rc, hst, st, res = T._entropy_decode_device(L, blobs)
        for k, e in enumerate(expect):
            def save(why):
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                fn = os.path.join(ROOT, "gpurun_out", "fuzz_input_gpu_%d_%d.jpg" % (b, k))
                open(fn, "wb").write(blobs[k])
                raise AssertionError("batch %d file %d (%s, %s; saved as %s): %s" % (b, k, os.environ["GAMUT_HIP_JPEG_PROGRESSIVE"], os.environ["GAMUT_HIP_JPEG_UNSTUFF"] or "auto", fn, why))
            if (hst[k] == 0 and st[k] == 0) != (e is not None):
                save("status %d / %d, the oracle %s" % (hst[k], int(st[k]), "decodes it" if e is not None else "refuses it"))
            if e is None:
                n_null += 1; continue
            if not (np.array_equal(res[k][0], e.coeffs) and np.array_equal(res[k][1], e.max_zag)):
                save("coefficients differ")
            if not O.same_density((res[k][2].pixel_aspect_ratio, res[k][2].dpi_y), (e.pixel_aspect_ratio, e.dpi_y)):
                save("density %r / %r, the oracle's %r / %r" % (res[k][2].pixel_aspect_ratio, res[k][2].dpi_y, e.pixel_aspect_ratio, e.dpi_y))
            n_img += 1
        comps = (4, 3, 1)[b % 3]
        rc, hst, px = T._decode_batch_device(L, blobs, comps)
        for k, e in enumerate(expect):
            if (hst[k] == 0) != (e is not None):
                raise AssertionError("batch %d file %d, files -> pixels: status %d, the oracle %s" % (b, k, hst[k], "decodes it" if e is not None else "refuses it"))
            if e is not None:
                want = O.decompress_jpeg(blobs[k], comps)[0]
                if not np.array_equal(px[k].reshape(-1), np.ascontiguousarray(want).reshape(-1)):
                    open(os.path.join(ROOT, "gpurun_out", "fuzz_input_gpu_px_%d_%d.jpg" % (b, k)), "wb").write(blobs[k])
                    raise AssertionError("batch %d file %d, files -> pixels: pixels differ" % (b, k))
        n_files += len(blobs)
    print("fuzz_input_gpu: %d batches, %d files (seed %d): %d decoded like the oracle (coefficients, max_zag, density; pixels through the files -> pixels call), "
          "%d refused by both; progressive scans on the device / the host, scans unstuffed on the device / the host / by size" % (batches, n_files, seed, n_img, n_null))


if __name__ == "__main__":
    main()
