cd $GRAFT_REPO_ROOT
python -m pytest tests/test_jpeg_gpu.py tests/test_image_gpu.py tests/test_oob_gpu.py -m gpu -x -q 2>&1 | tail -6
for wl in jpeg jpeg:4:1 jpeg:3:1 jpeg:1:0 jpeg:4:0 jpeg:4:2 jpeg:4:3 jpeg:3:4 jpeg:1:4; do
python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$wl', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['frac'], r['value'], r['config']['parity_check'][:40])
    else: print(l[:300])
"; done 2>&1 | tee gpurun_out/r04_jpeg_variants.txt
