cd $GRAFT_REPO_ROOT
python -m pytest tests/test_jpeg_gpu.py tests/test_qoi_gpu.py tests/test_batch_concurrency_gpu.py tests/test_png_gpu.py -m gpu -x -q 2>&1 | tail -4
for wl in jpeg:4:1 jpeg:3:1; do
python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$wl', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['frac'], r['config']['parity_check'][:40])
"; done
python tools/e2e_bench.py --batch 1024 --paths c --reps 6 2>&1 | grep -v amdgpu.ids
python tools/e2e_bench.py --batch 256 --paths c --reps 6 2>&1 | grep -v amdgpu.ids
python tools/e2e_mixed_bench.py --batch 3072 2>&1 | grep -v amdgpu.ids
python tools/files_bench.py 2>&1 | grep "^{" | cut -c1-330
