cd $GRAFT_REPO_ROOT
python -m pytest tests/test_png_gpu.py tests/test_batch_concurrency_gpu.py tests/test_stream_comm.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_t5.log
tail -4 gpurun_out/r04_t5.log
for wl in png png:heuristic; do
python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$wl', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['frac'], r['config']['parity_check'])
    else: print(l)
"; done 2>&1 | tee gpurun_out/r04_png_perband.txt
python tools/e2e_mixed_bench.py --batch 768 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_mixed_e2e.txt
python tools/e2e_mixed_bench.py --batch 3072 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04_mixed_e2e.txt
