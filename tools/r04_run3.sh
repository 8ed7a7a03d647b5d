cd $GRAFT_REPO_ROOT
python -m pytest tests/test_png_gpu.py tests/test_oob_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_t4_png.log
tail -3 gpurun_out/r04_t4_png.log
for al in 0 1; do for wl in png png:heuristic "png:random --width 1920 --height 1080 --batch 1024" "png:heuristic:3:3"; do
GAMUT_HIP_PNG_ALIGNED=$al python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('aligned=$al $wl', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['frac'], r['config']['parity_check'])
    else: print(l)
"; done; done 2>&1 | tee gpurun_out/r04_png_aligned2.txt
