#!/bin/bash
# A/B on one box: jprev = jpeg.hip of the commit before (nontemporal pixel stores always), base = this tree (4:2:0 -> rgba8: plain stores for rows off the 128-byte lines)
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
for rep in 1 2 3; do for v in jprev base; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for spec in "jpeg 1080 1920 1024" "jpeg 1366 768 2048" "jpeg 1000 1000 2048" "jpeg 1920 1080 1024"; do
    set -- $spec
    GAMUT_HIP_LIB=$L timeout 200 python bench.py --workload $1 --width $2 --height $3 --batch $4 --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v %-10s %5dx%-5d %8.4f ms  min %8.4f  frac %.4f  %s' % ('$1', $2, $3, r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'], r['roofline']['frac'], r['config']['parity_check'][:24]))
"
  done; done; done > gpurun_out/r06_jpeg_nt_ab.txt 2>&1
