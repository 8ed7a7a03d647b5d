timeout 600 python -m pytest tests/test_jpeg_gpu.py -m gpu -x -q -k "entropy or progressive" 2>&1 | tail -3
export GAMUT_HIP_TRACE=1
for B in 256 1024; do
timeout 600 python tools/e2e_bench.py --progressive --batch $B --reps 1 2>&1 | grep -v "amdgpu.ids" | tail -8
done
