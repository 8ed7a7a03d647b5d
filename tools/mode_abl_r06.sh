#!/bin/bash
# Round 6: which side of the PNG heuristic case carries the placement sensitivity?  The shipped library and its three ablations (fast tiles without
# their loads / stores / both: wrong pixels, timing only) on K fresh physical placements each (tools/png_mode_probe.py --only stride, plain strides).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for v in base rabl1 rabl2 rabl3; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  echo "== $v  (rabl1 = no prefetch loads, rabl2 = no write-back stores, rabl3 = neither)"
  PROBE_PLAIN=1 GAMUT_HIP_LIB=$L timeout 300 python tools/png_mode_probe.py --only stride --trials ${K:-8} 2>&1 | grep stride
done > gpurun_out/r06_png_mode_abl.txt 2>&1
