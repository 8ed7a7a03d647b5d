#!/usr/bin/env python3
"""Mutated progressive JPEG files through the device path of gamut_hip_jpeg_entropy_decode_device (jpeg_prog.hpp): the call must
return, nothing may fault (run tools/oob_probe.py for the placement-sensitive reads), intact files of the same batch must decode
to the oracle's coefficients, and a file the host feeder decodes without complaint must give the same coefficients on the GPU.
Usage: python tools/fuzz_prog_gpu.py [batches=20] [baseline]     (baseline: sequential files, large enough for the self-synchronising kernel)"""
import ctypes as C
import io
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GAMUT_HIP_JPEG_PROGRESSIVE"] = "device"
import gen  # noqa: E402
import torch  # noqa: E402,F401
from gamut_amd import _capi  # noqa: E402


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    progressive = not (len(sys.argv) > 2 and sys.argv[2] == "baseline")
    L = _capi.lib()
    _capi.check(L.gamut_hip_init(0))
    rng = np.random.default_rng(9)
    seeds = []
    for k, kw in enumerate((dict(quality=90, subsampling=2), dict(quality=60, subsampling=0, optimize=True), dict(quality=95, subsampling=1, restart_marker_blocks=4),
                            dict(quality=40, subsampling=2, restart_marker_rows=1))):
        dims = (176 + 8 * k, 120 - 8 * k) if progressive else (640 + 16 * k, 480 - 16 * k)
        bio = io.BytesIO(); Image.fromarray(gen.synth_rgb(dims[0], dims[1], 70 + k)).save(bio, "JPEG", progressive=progressive, **kw); seeds.append(bio.getvalue())
    bio = io.BytesIO(); Image.fromarray(gen.synth_rgb(97 if progressive else 600, 61 if progressive else 400, 80)).convert("L").save(bio, "JPEG", progressive=progressive, quality=80); seeds.append(bio.getvalue())
    if progressive:                                             # scan scripts libjpeg does not write (tests/jpeg_scripts.py): scans that do not line up take the per-file barrier
        import jpeg_scripts as J
        bio = io.BytesIO(); Image.fromarray(gen.synth_rgb(168, 104, 81)).save(bio, "JPEG", quality=85, subsampling=2)
        seeds += [J.progressive_with_script(bio.getvalue(), J.FOUR_BANDS, 0), J.progressive_with_script(bio.getvalue(), J.DEEP, 7)]
    n_same = n_flag = n_rej = 0
    for b in range(batches):
        blobs = []
        for k in range(24):
            data = bytearray(seeds[(b + k) % len(seeds)])
            if k % 6:                                           # every sixth file stays intact
                sos = data.index(b"\xff\xda")
                for _ in range(int(rng.integers(1, 5))):
                    kind = int(rng.integers(0, 4))
                    if len(data) < 8:
                        break                                   # (an earlier truncation left nothing to mutate)
                    i = int(rng.integers(min(sos, len(data) - 2) if kind < 3 else 2, len(data) - 1))
                    if kind == 0:
                        data[i] = int(rng.integers(0, 256))
                    elif kind == 1:
                        data = data[:i]
                    elif kind == 2:
                        data[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
                    else:
                        data[i] = int(rng.integers(0, 256))
            blobs.append(bytes(data))
        n = len(blobs)
        bufs = [np.frombuffer(x, np.uint8) if len(x) else np.zeros(1, np.uint8) for x in blobs]
        ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in bufs]); lens = (C.c_size_t * n)(*[len(x) for x in blobs])
        hdr = (_capi.JpegFrame * n)(); nblk = []
        for i in range(n):
            rc = L.gamut_hip_jpeg_read_header(ptrs[i], lens[i], C.byref(hdr[i]))
            nblk.append(hdr[i].mcus_per_row * hdr[i].mcus_per_col * hdr[i].blocks_per_mcu if rc == 0 else 0)
        if sum(nblk) > 4_000_000:                               # a mutated SOF asked for a huge frame: not this tool's subject
            continue
        off = np.concatenate([[0], np.cumsum(nblk)[:-1]]).astype(np.int64)
        co_off = off * 64
        total = max(1, sum(nblk))
        dco = torch.full((total * 64,), 0x5A5A, dtype=torch.int16, device="cuda"); dzz = torch.full((total,), 0xEE, dtype=torch.uint8, device="cuda")
        dst = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        info = (_capi.JpegFrame * n)(); hst = (C.c_int * n)()
        L.gamut_hip_jpeg_entropy_decode_device(ptrs, lens, n, co_off.ctypes.data_as(C.POINTER(C.c_int64)), off.ctypes.data_as(C.POINTER(C.c_int64)),
                                               dco.data_ptr(), dzz.data_ptr(), dst.data_ptr(), info, hst, None)
        torch.cuda.synchronize()
        co = dco.cpu().numpy(); zz = dzz.cpu().numpy(); st = dst.cpu().numpy()
        for i in range(n):
            if hst[i] != 0:
                n_rej += 1
                continue
            fr = _capi.JpegFrame()
            rc = L.gamut_hip_jpeg_decode_coeffs(ptrs[i], lens[i], C.byref(fr))          # the host feeder on the same bytes
            if rc == 0 and st[i] == 0:
                ref = np.ctypeslib.as_array(fr.coeffs, (nblk[i] * 64,)); refz = np.ctypeslib.as_array(fr.max_zag, (nblk[i],))
                if not (np.array_equal(co[co_off[i]:co_off[i] + nblk[i] * 64], ref) and np.array_equal(zz[off[i]:off[i] + nblk[i]], refz)):
                    got = co[co_off[i]:co_off[i] + nblk[i] * 64].reshape(-1, 64); want = ref.reshape(-1, 64)
                    bad = np.argwhere(got != want)
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    open(os.path.join(ROOT, "gpurun_out", f"fuzz_prog_mismatch_{b}_{i}.jpg"), "wb").write(blobs[i])
                    raise AssertionError(f"batch {b} file {i} (seed file {(b + i) % len(seeds)}, {len(blobs[i])} bytes, saved): {len(bad)} coefficients differ, first at "
                                         f"{bad[:4].tolist()}: device {[int(got[x, y]) for x, y in bad[:4]]} host {[int(want[x, y]) for x, y in bad[:4]]}; "
                                         f"max_zag differs in {int((zz[off[i]:off[i] + nblk[i]] != refz).sum())} blocks")
                n_same += 1
            else:
                assert i % 6 != 0, "an intact file was flagged"
                n_flag += 1
            L.gamut_hip_jpeg_frame_free(C.byref(fr))
    print(f"fuzz_prog_gpu ({'progressive' if progressive else 'baseline'}): {batches} batches of 24 files: {n_same} decoded like the host feeder, {n_flag} flagged by either decoder, {n_rej} rejected by the marker walk")


if __name__ == "__main__":
    main()
