cd $GRAFT_REPO_ROOT
python -m pytest tests/test_png_gpu.py tests/test_oob_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_t2_png.log
python -m pytest tests/test_jpeg_gpu.py tests/test_stream_comm.py tests/test_image_gpu.py tests/test_bench_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04_t2_jpeg.log
for al in 0 1; do for wl in png png:heuristic; do
GAMUT_HIP_PNG_ALIGNED=$al python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-traffic --no-also 2>&1 | tail -1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('aligned=$al $wl', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['frac'], r['config']['parity_check'])
    else: print(l)
"; done; done > gpurun_out/r04_png_aligned.txt 2>&1
GAMUT_HIP_PNG_ALIGNED=1 QUEUES=1 bash tools/png_reads.sh base 512 >> gpurun_out/r04_png_aligned.txt 2>&1
python tools/files_bench.py jpeg > gpurun_out/r04_files_jpeg.txt 2>&1
tail -5 gpurun_out/r04_t2_png.log gpurun_out/r04_t2_jpeg.log; cat gpurun_out/r04_png_aligned.txt gpurun_out/r04_files_jpeg.txt | cut -c1-400
