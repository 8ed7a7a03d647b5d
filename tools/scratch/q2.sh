run() { # env... -- workload
  timeout 300 python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$TAG', '$WL', r['value'], 'Mpx/s', r['roofline']['achieved'], 'GB/s')
"
}
for WL in "png" "png:random" "png --width 1920 --height 1080 --batch 341" "png:random --width 1920 --height 1080 --batch 341" "png --batch 64"; do
  TAG="q0" GAMUT_HIP_PNG_QUEUE=0 run
  TAG="q1,group=count" GAMUT_HIP_PNG_QUEUE=1 run
  TAG="q1,group=128" GAMUT_HIP_PNG_QUEUE=1 GAMUT_HIP_PNG_GROUP=128 run
done
