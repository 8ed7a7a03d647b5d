#!/bin/bash
# Run-to-run spread of the convert kernels: K fresh processes per directed pair of config 4 on ONE box, the kernel's average and
# minimum launch (HIP events inside bench.py, after its untimed pre-warm) -> gpurun_out/r06_convert_run_to_run.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
K=${1:-5}
for p in $(seq 1 $K); do for w in convert:rgbaf32:rgba8 convert:rgba8:rgba16 convert:rgba16:rgba8 convert:rgba16:rgbaf32; do
  timeout 300 python bench.py --workload $w --batch 64 --steps 10 --warmup 2 --no-cpu --no-traffic --no-also 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('process $p %-24s %8.4f ms  min %8.4f  frac %.4f' % ('$w', r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'], r['roofline']['frac']))"
done; done > gpurun_out/r06_convert_run_to_run.txt 2>&1
