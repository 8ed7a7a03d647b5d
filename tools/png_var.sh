# bash tools/png_var.sh "<variants>" : bench.py png workloads over library variants built by tools/variant.sh
for v in ${1:-base}; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for wl in "png" "png:random" "png:heuristic:3:4" "png --width 1920 --height 1080 --batch 1024" "png:random --width 1920 --height 1080 --batch 341"; do
    GAMUT_HIP_LIB=$L timeout 200 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v', '$wl', r['value'], 'Mpx/s', r['roofline']['achieved'], 'GB/s', r['roofline']['kernel_ms_avg'], 'ms')
"
  done
done
