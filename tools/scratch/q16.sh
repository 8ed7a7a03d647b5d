timeout 600 python tools/fuzz_prog_gpu.py 25 2>&1 | grep -v amdgpu.ids | tail -12
