#!/usr/bin/env python3
"""png_mode_probe -- what selects the slow mode of the PNG de-filter launch?

The heuristic-filter case of BASELINE.json configs[2] (512 x 3840x2160 RGBA8, no Paeth rows: the line-aligned, HBM-bound path)
ran 6.05 ms in one process and 6.8 ms in the next on the same box, constant inside a process.  This script times the SAME filtered
streams in ONE process while only the placement of the two big buffers changes:

  initial            raw / out as torch's caching allocator hands them out (what bench.py does)
  out#k              `out` freed back to the driver (empty_cache) and allocated again
  raw#k              the streams copied into a fresh allocation, the old one freed
  pad#k              a block of odd size allocated (and kept) in front of fresh raw + out: shifts both
  hip#k              raw / out from hipMalloc directly (no torch pool), hipMemcpy of the streams
  arena+off          both buffers inside one hipMalloc block, `out` at raw_end rounded up to 2 MiB + off

Per trial: virtual addresses, average / minimum of --steps launches (HIP events).  Usage:
    python tools/png_mode_probe.py [--policy heuristic|random] [--batch 512] [--steps 10] [--trials 3]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gamut_amd import _capi, synth          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--policy", default="heuristic")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--only", default="", help="comma list of trial families (initial,out,raw,pad,hip,arena); default all but stride")
    a = ap.parse_args()
    fams = set(a.only.split(",")) if a.only else {"initial", "out", "raw", "pad", "hip", "arena"}
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L = _capi.lib()
    _capi.check(L.gamut_hip_init(0))
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    stream = torch.cuda.current_stream().cuda_stream
    w, h, B = a.width, 2160, a.batch
    policy = int(a.policy) if a.policy.isdigit() else a.policy
    raw, _ = synth.png_raw_batch(B, w, h, dev, seed=3, policy=policy, channels=4)
    raw_len = raw.shape[1]
    out_len = w * h * 4
    status = torch.zeros((B,), dtype=torch.int32, device=dev)
    algo = B * (raw_len + out_len)

    def launch(rp, op):
        _capi.check(L.gamut_hip_png_defilter_batch_device(rp, raw_len, raw_len, op, out_len, w, h, 4, 4, 8, 6, B, status.data_ptr(), stream))

    def measure(tag, rp, op):
        for _ in range(3):
            launch(rp, op)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        for s, e in ev:
            s.record(); launch(rp, op); e.record()
        torch.cuda.synchronize()
        ms = [s.elapsed_time(e) for s, e in ev]
        avg = sum(ms) / len(ms)
        print(f"{tag:14s} raw {rp:#016x} out {op:#016x}  (raw % 2M = {rp % (2 << 20):#9x}, out % 2M = {op % (2 << 20):#9x})  "
              f"avg {avg:.3f} ms  min {min(ms):.3f}  frac {algo / avg / 1e6 / 8000:.3f}", flush=True)
        return avg

    out = torch.empty((B, out_len), dtype=torch.uint8, device=dev)
    if "initial" in fams:
        for k in range(2):
            measure(f"initial#{k}", raw.data_ptr(), out.data_ptr())
    if "out" in fams:
        for k in range(a.trials):
            del out
            torch.cuda.empty_cache()
            out = torch.empty((B, out_len), dtype=torch.uint8, device=dev)
            measure(f"out#{k}", raw.data_ptr(), out.data_ptr())
    if "raw" in fams:
        for k in range(a.trials):
            raw2 = raw.clone()
            del raw
            torch.cuda.empty_cache()
            raw = raw2; del raw2
            measure(f"raw#{k}", raw.data_ptr(), out.data_ptr())
    pads = []
    if "pad" in fams:
        for k, sz in enumerate([(1 << 20) + 4096, (3 << 20) + 12288, (1 << 30) + (5 << 12), 777 << 12][:max(a.trials, 1) + 1]):
            pads.append(torch.empty(sz, dtype=torch.uint8, device=dev))
            raw2 = raw.clone(); del raw; del out
            torch.cuda.empty_cache()
            raw = raw2; del raw2
            out = torch.empty((B, out_len), dtype=torch.uint8, device=dev)
            measure(f"pad#{k}", raw.data_ptr(), out.data_ptr())
    if "hip" in fams or "arena" in fams or "stride" in fams:
        del out
        torch.cuda.empty_cache()
    if "hip" in fams:
        for k in range(a.trials):
            pr, po = C.c_void_p(), C.c_void_p()
            assert hip.hipMalloc(C.byref(pr), B * raw_len) == 0 and hip.hipMalloc(C.byref(po), B * out_len) == 0
            assert hip.hipMemcpy(pr, C.c_void_p(raw.data_ptr()), B * raw_len, 3) == 0
            measure(f"hip#{k}", pr.value, po.value)
            hip.hipFree(pr); hip.hipFree(po)
    if "arena" in fams:
        M2 = 2 << 20
        rsz = (B * raw_len + M2 - 1) // M2 * M2
        for off in [0, 4096, 65536, 1 << 20, M2 + 256, 3 * 64 * 1024 + 128][:a.trials + 3]:
            pa = C.c_void_p()
            assert hip.hipMalloc(C.byref(pa), rsz + B * out_len + 8 * M2) == 0
            assert hip.hipMemcpy(pa, C.c_void_p(raw.data_ptr()), B * raw_len, 3) == 0
            measure(f"arena+{off:#x}", pa.value, pa.value + rsz + off)
            hip.hipFree(pa)
    if "stride" in fams:
        # Is it the LOW address bits the images share?  Images laid out at strides that are multiples of 8 KiB (out: 3840 x 2160 x 4) put
        # every concurrently running wave -- all at the same x of their rows -- on the same address bits below the image number; pads
        # break that.  Per (raw pad, out pad): K fresh hipMalloc placements, every one timed.
        import statistics
        K = max(a.trials, 3)
        for rpad, opad in ([(0, 0), (0, 256), (0, 1280), (0, 4096 + 256), (0, 65536 + 1280), (128, 0), (128, 1280), (4096 + 128, 65536 + 1280), (0, 33 * 1024), (0, 2 << 20)] if not os.environ.get('PROBE_PLAIN') else [(0, 0)]):
            ts = []
            for k in range(K):
                rs, os_ = raw_len + rpad, out_len + opad
                pr, po = C.c_void_p(), C.c_void_p()
                assert hip.hipMalloc(C.byref(pr), B * rs + 256) == 0 and hip.hipMalloc(C.byref(po), B * os_ + 256) == 0
                assert hip.hipMemcpy2D(pr, rs, C.c_void_p(raw.data_ptr()), raw_len, raw_len, B, 3) == 0

                def launch2(rp=pr.value, op=po.value, rs=rs, os_=os_):
                    _capi.check(L.gamut_hip_png_defilter_batch_device(rp, rs, raw_len, op, os_, w, h, 4, 4, 8, 6, B, status.data_ptr(), stream))
                for _ in range(3):
                    launch2()
                torch.cuda.synchronize()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
                for s_, e_ in ev:
                    s_.record(); launch2(); e_.record()
                torch.cuda.synchronize()
                ts.append(sum(s_.elapsed_time(e_) for s_, e_ in ev) / len(ev))
                hip.hipFree(pr); hip.hipFree(po)
            print(f"w {w} stride raw+{rpad:<6d} out+{opad:<8d}: " + " ".join(f"{t:.3f}" for t in ts) + f"   min {min(ts):.3f} median {statistics.median(ts):.3f} max {max(ts):.3f}", flush=True)
    assert int(status.abs().sum()) == 0


if __name__ == "__main__":
    main()
