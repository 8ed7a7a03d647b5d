timeout 600 python -m pytest tests/test_jpeg_gpu.py tests/test_batch_concurrency_gpu.py tests/test_oob_gpu.py -m gpu -x -q 2>&1 | tail -3
GAMUT_HIP_TRACE=1 timeout 600 python tools/e2e_bench.py --batch 1024 --reps 2 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python tools/e2e_bench.py --batch 256 --reps 2 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python tools/e2e_bench.py --batch 1024 --reps 2 --restart-rows 1 2>&1 | grep -v amdgpu.ids | tail -2
