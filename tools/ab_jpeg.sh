# bash tools/ab_jpeg.sh "<variants>" ["<workloads>"] : A/B of library variants (tools/variant.sh) on one box, interleaved repetitions
for rep in 1 2 3; do
for v in ${1:-base}; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for wl in ${2:-jpeg}; do
  GAMUT_HIP_LIB=$L GAMUT_BENCH_NOCHECK=${NOCHECK:-} timeout 200 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v $wl', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])
"
  done
done; done
