for rep in 1 2; do
for v in base jntl; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for wl in jpeg jpeg:3 jpeg:4:1; do
  GAMUT_HIP_LIB=$L timeout 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v $wl', r['value'], 'Mpx/s', r['roofline']['achieved'], 'GB/s', r['roofline']['kernel_ms_avg'], 'ms')
"
  done
done
for v in base pntl; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for wl in png png:random png:heuristic:3; do
  GAMUT_HIP_LIB=$L timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v $wl', r['value'], 'Mpx/s', r['roofline']['achieved'], 'GB/s', r['roofline']['kernel_ms_avg'], 'ms')
"
  done
done
done
bash tools/conv_var.sh base "rgba16:rgbaf32 rgbaf32:rgba8 rgba8:rgba16 rgba16:rgba8 rgbaf32:rgba16 rgba8:rgbaf32"
