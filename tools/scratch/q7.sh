timeout 600 python -m pytest tests/test_qoi_gpu.py tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -4
for extra in "--serial-formats" ""; do
  timeout 300 python bench.py --workload mixed --steps 5 --warmup 1 --no-cpu $extra 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('mixed $extra', r['value'], 'Mpx/s', r['ms_per_step'], 'ms', {k:v['ms'] for k,v in r['config']['per_format'].items()})"
done
run() {
  timeout 300 python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$TAG', r['config']['workload'][:60], r['value'], 'Mpx/s', r['roofline']['achieved'], 'GB/s')
"
}
for WL in "png --width 1920 --height 1080 --batch 341" "png:random --width 1920 --height 1080 --batch 341" "png --width 1920 --height 1080 --batch 1024" "png --batch 256" "png:random --batch 256" "png --batch 128"; do
  TAG="q0" GAMUT_HIP_PNG_QUEUE=0 run
  TAG="q1" GAMUT_HIP_PNG_QUEUE=1 run
  TAG="auto" run
done
