timeout 900 python -m pytest tests/test_png_gpu.py tests/test_oob_gpu.py -m gpu -x -q 2>&1 | tail -4
run() {
  timeout 300 python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu --no-traffic 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$TAG', r['config']['workload'][:70], r['value'], 'Mpx/s', r['roofline']['achieved'], 'GB/s')
"
}
for WL in "png --batch 512" "png:random --batch 512" "png:4 --batch 512" "png:random:3 --batch 512" "png:random:3:4 --batch 512" "png:random --batch 341 --width 1920 --height 1080" "png --batch 341 --width 1920 --height 1080"; do
  TAG="q0" GAMUT_HIP_PNG_QUEUE=0 run
  TAG="q1" GAMUT_HIP_PNG_QUEUE=1 run
done
