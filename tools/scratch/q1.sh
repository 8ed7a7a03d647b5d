export GAMUT_HIP_PNG_QUEUE=1
timeout 900 python -m pytest tests/test_png_gpu.py tests/test_oob_gpu.py -m gpu -x -q 2>&1 | tail -8
unset GAMUT_HIP_PNG_QUEUE
for q in 0 1; do
 for wl in "png" "png:random" "png --width 1920 --height 1080 --batch 341" "png:random --width 1920 --height 1080 --batch 341" "png:heuristic:3:4" "png --batch 64" "png --batch 8" "png --batch 1"; do
  GAMUT_HIP_PNG_QUEUE=$q timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('queue=$q', '$wl', r['value'], 'Mpx/s', r['roofline']['achieved'], 'GB/s', r['roofline']['kernel_ms_avg'], 'ms')
"
 done
done
