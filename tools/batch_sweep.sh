#!/bin/bash
# Does a workload's time per image depend on the batch?  bench.py at several batch sizes on one box (kernel average per image) -> gpurun_out/r06_batch_sweep.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
for spec in "jpeg 256" "jpeg 512" "jpeg 1024" "jpeg 2048" "jpeg 4096" "png 128" "png 256" "png 512" "png 1024" "png:heuristic 128" "png:heuristic 256" "png:heuristic 512" "png:heuristic 1024"; do
  set -- $spec
  GAMUT_BENCH_NOCHECK=1 timeout 300 python bench.py --workload $1 --batch $2 --steps 15 --warmup 3 --no-cpu --no-traffic --no-also 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('%-14s batch %5d  %8.4f ms  %8.3f us per image  frac %.4f' % ('$1', $2, r['roofline']['kernel_ms_avg'], 1000*r['roofline']['kernel_ms_avg']/$2, r['roofline']['frac']))"
done > gpurun_out/r06_batch_sweep.txt 2>&1
