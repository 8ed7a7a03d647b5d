# bash tools/conv_var.sh "<variants>" "<pairs>" : bench.py convert workloads over library variants built by tools/variant.sh
for v in ${1:-base}; do
  for wl in ${2:-rgba16:rgbaf32 rgbaf32:rgba8 rgba8:rgba16 rgba8:rgbaf32}; do
    if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
    GAMUT_HIP_LIB=$L timeout 120 python bench.py --workload convert:$wl --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v', '$wl', r['roofline']['achieved'], 'GB/s', r['roofline']['kernel_ms_avg'], 'ms')
"
  done
done
