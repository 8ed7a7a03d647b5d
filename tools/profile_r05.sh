#!/bin/bash
# Round 5's profile set, on the GPU box:  bash tools/profile_r05.sh   -> gpurun_out/summary_r05_*/ and gpurun_out/r05_*.txt (copied into profiles/)
# Kernel statistics are those of the TIMED dispatches of each profiled bench.py run (tools/summarize_prof.py); the PMC passes (FETCH_SIZE /
# WRITE_SIZE) run at the TIMED batch of every workload (1024 JPEG images, 512 PNG images).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile.sh r05_jpeg 1024 -- --steps 30 --warmup 10
bash tools/profile.sh r05_jpeg_photo 1024 -- --workload jpeg:photo --steps 30 --warmup 10
bash tools/profile.sh r05_jpeg_444 1024 -- --workload jpeg:4:1 --steps 20 --warmup 5
bash tools/profile.sh r05_jpeg_rgb8 1024 -- --workload jpeg:3 --steps 20 --warmup 5
bash tools/profile.sh r05_jpeg_l8 1024 -- --workload jpeg:1 --steps 20 --warmup 5
bash tools/profile.sh r05_jpeg_422 1024 -- --workload jpeg:4:2 --steps 20 --warmup 5
bash tools/profile.sh r05_jpeg_440 1024 -- --workload jpeg:4:3 --steps 20 --warmup 5
bash tools/profile.sh r05_png_random 512 -- --workload png --steps 10 --warmup 3
bash tools/profile.sh r05_png 512 -- --workload png:heuristic --steps 10 --warmup 3
PER_STEP=4 bash tools/profile.sh r05_convert_rgba16_rgbaf32 2 -- --workload convert:rgba16:rgbaf32 --batch 256 --steps 6 --warmup 2
PER_STEP=3 bash tools/profile.sh r05_convert_rgbaf32_rgba8 2 -- --workload convert:rgbaf32:rgba8 --batch 256 --steps 6 --warmup 2
PER_STEP=2 bash tools/profile.sh r05_convert_rgba8_rgba16 2 -- --workload convert:rgba8:rgba16 --batch 256 --steps 6 --warmup 2
# the mixed step at config 5's size on one GPU, the file-level feeders: kernel trace only
trace() {   # trace <tag> <timed dispatches per kernel> -- <command...>
  local tag=$1 steps=$2; shift 3
  mkdir -p $R/gpurun_out/summary_$tag
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$tag && mkdir -p /tmp/tr_$tag &&
   timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/tr_$tag/trace -o t -- "$@" > $R/gpurun_out/summary_$tag/run.txt 2>&1;
   python $R/tools/summarize_prof.py /tmp/tr_$tag $R/gpurun_out/summary_$tag 0 $steps | head -12; rm -rf /tmp/tr_$tag)
}
trace r05_mixed_8192 10 -- python $R/bench.py --workload mixed --total-images 8192 --steps 10 --warmup 3 --no-cpu --no-traffic --no-also
trace r05_jpeg_files 0 -- python $R/tools/e2e_bench.py --batch 1024 --paths c --reps 4
trace r05_progressive_files 0 -- python $R/tools/e2e_bench.py --batch 1024 --paths c --reps 2 --progressive
trace r05_png_files 0 -- python $R/tools/e2e_png_bench.py --batch 256
# what a caller with files sees; the same with two host threads (a rank's share on an 8-rank node)
python tools/files_bench.py > gpurun_out/r05_files_bench.jsonl 2>/dev/null
GAMUT_HIP_HOST_THREADS=2 python tools/files_bench.py > gpurun_out/r05_files_bench_2threads.jsonl 2>/dev/null
python tools/e2e_mixed_bench.py --batch 768 > gpurun_out/r05_mixed_e2e.txt 2>&1; python tools/e2e_mixed_bench.py --batch 3072 >> gpurun_out/r05_mixed_e2e.txt 2>&1
(time python bench.py) > gpurun_out/r05_bench_default.log 2>&1
