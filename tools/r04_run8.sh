cd $GRAFT_REPO_ROOT
for v in base m10 m8 m12 m10nt0 base m10; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for wl in jpeg:4:1 jpeg:3:1; do
  GAMUT_BENCH_NOCHECK=1 GAMUT_HIP_LIB=$L python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v $wl', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['frac'])
"; done; done 2>&1 | tee gpurun_out/r04_jpeg_444_var.txt
