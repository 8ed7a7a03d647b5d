for rep in 1 2 3 4; do
for v in base old; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  GAMUT_HIP_LIB=$L timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v', r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'])
"
done; done
