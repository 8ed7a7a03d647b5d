cd $GRAFT_REPO_ROOT
python -m pytest tests/test_jpeg_gpu.py tests/test_qoi_gpu.py tests/test_batch_concurrency_gpu.py -m gpu -x -q 2>&1 | tail -4
for pin in "" "--pinned"; do
python tools/e2e_bench.py --batch 1024 --paths c --reps 6 $pin 2>&1 | grep -v amdgpu.ids
python tools/e2e_mixed_bench.py --batch 3072 $pin 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/r04_pinned.txt
