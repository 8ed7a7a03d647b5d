bash tools/profile.sh r02_png 64 -- --workload png --steps 10 --warmup 2 > /dev/null 2>&1
bash tools/profile.sh r02_png_random 64 -- --workload png:random --steps 10 --warmup 2 > /dev/null 2>&1
bash tools/profile.sh r02_png_rgb8_rgba8 64 -- --workload png:heuristic:3:4 --steps 10 --warmup 2 > /dev/null 2>&1
bash tools/profile.sh r02_jpeg 64 -- --steps 20 --warmup 5 > /dev/null 2>&1
bash tools/bench_all.sh > gpurun_out/bench_matrix.txt 2>&1
bash tools/pmc_valu.sh > gpurun_out/r02_pmc_valu.txt 2>&1
