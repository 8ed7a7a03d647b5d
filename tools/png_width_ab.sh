#!/bin/bash
# A/B on one box: prev = png.hip of round 5 (write-back groups = the rows' own pieces), base = this tree (groups = the lines of memory for rows off them)
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
for rep in 1 2; do for v in prev base; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for spec in "png:heuristic 3848 2160 512" "png:random 3848 2160 512" "png:heuristic 1080 1920 1024" "png:random 1080 1920 1024" "png:random 1000 1000 2048" "png:random:3:4 1080 1920 1024" "png:heuristic 3840 2160 512"; do
    set -- $spec
    GAMUT_HIP_LIB=$L timeout 200 python bench.py --workload $1 --width $2 --height $3 --batch $4 --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v $1 $2x$3', r['roofline']['kernel_ms_avg'], 'min', r['roofline']['kernel_ms_min'], r['roofline']['frac'], r['config']['parity_check'][:30])
"
  done; done; done > gpurun_out/r06_png_width_ab.txt 2>&1
