#!/bin/bash
# VERDICT r05 item 1 asked for "heuristic >= 0.68 in 10 of 10 consecutive processes": ten processes in a row, each its own draw of the physical
# placement.  Kernel average / minimum / fraction per process; the random-filter case and the headline beside it.
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
for i in $(seq 1 10); do
  for wl in png:heuristic png jpeg; do
    timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('process $i %-14s %8.4f ms  min %8.4f  frac %.4f' % ('$wl', r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'], r['roofline']['frac']))
"
  done
done > gpurun_out/r06_run_to_run.txt 2>&1
