#!/bin/bash
# Round 3's profile set, on the GPU box:  bash tools/profile_r03.sh   -> gpurun_out/summary_r03_*/ (copy into profiles/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile.sh r03_jpeg 64 -- --steps 30 --warmup 10
bash tools/profile.sh r03_jpeg_photo 64 -- --workload jpeg:photo --steps 30 --warmup 10
bash tools/profile.sh r03_png_random 16 -- --workload png --steps 10 --warmup 3
bash tools/profile.sh r03_png 16 -- --workload png:heuristic --steps 10 --warmup 3
# QOI (2730 streams, config 5's share on one GPU) and the mixed step: kernel trace only
mkdir -p gpurun_out/summary_r03_qoi
(cd /tmp && export TMPDIR=/tmp && QOI_BENCH_B=256,2730 timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/qoi_tr -o t -- python $R/tools/qoi_bench.py > $R/gpurun_out/summary_r03_qoi/qoi_bench.txt 2>&1; cp $(find /tmp/qoi_tr -name "*kernel_stats.csv" | head -1) $R/gpurun_out/summary_r03_qoi/kernel_stats.csv; rm -rf /tmp/qoi_tr)
python bench.py --workload mixed --total-images 8192 --steps 10 --warmup 3 --no-cpu > gpurun_out/summary_r03_qoi/mixed_8192_bench.json 2>/dev/null
bash tools/pmc_valu.sh > gpurun_out/summary_r03_pmc_valu.txt 2>&1
(time python bench.py) > gpurun_out/r03_bench_default.log 2>&1
# the feeders: the inflate kernel's own set, files -> pixels for the three formats
bash tools/inflate_profile.sh > gpurun_out/inflate_profile.log 2>&1
python tools/files_bench.py > gpurun_out/r03_files_bench.jsonl 2>/dev/null
python tools/e2e_mixed_bench.py --batch 768 > gpurun_out/r03_mixed_e2e.txt 2>&1; python tools/e2e_mixed_bench.py --batch 3072 >> gpurun_out/r03_mixed_e2e.txt 2>&1
