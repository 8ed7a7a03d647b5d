cd $GRAFT_REPO_ROOT
python -m pytest tests/test_png_gpu.py tests/test_oob_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_t3_png.log
tail -3 gpurun_out/r04_t3_png.log
for al in 0 1; do for wl in png png:heuristic; do
GAMUT_HIP_PNG_ALIGNED=$al python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('aligned=$al $wl', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['frac'], r['config']['parity_check'])
    else: print(l)
"; done; done 2>&1 | tee gpurun_out/r04_png_aligned.txt
for site in host device; do for g in 2 4 8; do
echo "== unstuff=$site groups=$g"
GAMUT_HIP_JPEG_UNSTUFF=$site GAMUT_HIP_JPEG_GROUPS=$g python tools/e2e_bench.py --batch 1024 --paths c --reps 6 2>&1 | grep "files ->"
done; done 2>&1 | tee gpurun_out/r04_jpeg_sweep.txt
GAMUT_HIP_TRACE=1 GAMUT_HIP_JPEG_UNSTUFF=device python tools/e2e_bench.py --batch 1024 --paths c --reps 2 2>&1 | grep gamut_hip | tail -2 | tee -a gpurun_out/r04_jpeg_sweep.txt
GAMUT_HIP_TRACE=1 GAMUT_HIP_JPEG_UNSTUFF=host python tools/e2e_bench.py --batch 1024 --paths c --reps 2 2>&1 | grep gamut_hip | tail -2 | tee -a gpurun_out/r04_jpeg_sweep.txt
