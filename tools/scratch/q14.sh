timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
export GAMUT_HIP_TRACE=1
timeout 900 python tools/e2e_bench.py --progressive --batch 4096 --reps 1 2>&1 | grep -v "amdgpu.ids" | tail -8
