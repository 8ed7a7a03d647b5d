for S in 256 512 1024; do timeout 600 python tools/inflate_bench.py --streams $S --reps 2 2>&1 | grep -v amdgpu.ids | tail -6; done
