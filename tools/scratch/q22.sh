run() {
  timeout 300 python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu --no-traffic 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$TAG', r['config']['workload'][:70], r['value'], 'Mpx/s', r['roofline']['achieved'], 'GB/s')
"
}
for WL in "png" "png:random" "png:random --batch 341 --width 1920 --height 1080"; do
  TAG="base" run
  for v in w9 ntl w6; do TAG=$v GAMUT_HIP_LIB=gamut_amd/lib/var/libgamut_hip_$v.so run; done
done
