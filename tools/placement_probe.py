#!/usr/bin/env python3
"""placement_probe -- does the PHYSICAL placement of two 17 GB buffers move a plain streaming kernel the way it moves the PNG
de-filter's HBM-bound case (tools/png_mode_probe.py: same virtual addresses, fresh physical pages, 6.03 ... 6.96 ms)?

K times: free both buffers back to the driver, allocate them again (hipMalloc), and time
  copy   dst <- src, hipMemcpyAsync device to device            (the runtime's own copy kernel: read + write)
  fill   hipMemsetAsync(dst)                                    (write only)
  conv   gamut_hip_scanlines_convert_device rgba8 -> rgba8      (this library's streaming kernel: 16 bytes per lane, read + write)
  read   torch sum over src as int32                             (read only)
Prints one line per placement and the spread per kernel."""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gamut_amd import _capi          # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nbytes = int(float(sys.argv[2]) * (1 << 30)) if len(sys.argv) > 2 else 16 << 30
    torch.cuda.set_device(0)
    L = _capi.lib()
    _capi.check(L.gamut_hip_init(0))
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    stream = torch.cuda.current_stream().cuda_stream
    w = 8192
    h = nbytes // (w * 4)
    res = {"copy": [], "fill": [], "conv": []}

    def timed(fn, reps=6):
        fn(); fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for s, e in ev:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        return sum(s.elapsed_time(e) for s, e in ev) / reps

    for k in range(K):
        ps, pd = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(ps), nbytes) == 0 and hip.hipMalloc(C.byref(pd), nbytes) == 0
        hip.hipMemsetAsync(ps, 0x5a, nbytes, stream)
        t_copy = timed(lambda: hip.hipMemcpyAsync(pd, ps, nbytes, 3, stream))
        t_fill = timed(lambda: hip.hipMemsetAsync(pd, 1, nbytes, stream))
        t_conv = timed(lambda: _capi.check(L.gamut_hip_scanlines_convert_device(12, ps.value, w * 4, 0, 12, pd.value, w * 4, 0, w, h, 1, stream)))
        res["copy"].append(t_copy); res["fill"].append(t_fill); res["conv"].append(t_conv)
        print(f"placement {k}: src {ps.value:#x} dst {pd.value:#x}  copy {t_copy:.3f} ms ({2 * nbytes / t_copy / 1e9:.2f} TB/s)  fill {t_fill:.3f} ms ({nbytes / t_fill / 1e9:.2f} TB/s)  "
              f"convert rgba8->rgba8 {t_conv:.3f} ms ({2 * nbytes / t_conv / 1e9:.2f} TB/s)", flush=True)
        hip.hipFree(ps); hip.hipFree(pd)
    for name, ts in res.items():
        print(f"{name}: min {min(ts):.3f} median {statistics.median(ts):.3f} max {max(ts):.3f} ms  spread {100 * (max(ts) / min(ts) - 1):.1f} %")


if __name__ == "__main__":
    main()
