#!/bin/bash
# Round 4's profile set, on the GPU box:  bash tools/profile_r04.sh   -> gpurun_out/summary_r04_*/ and gpurun_out/r04_*.txt (copied into profiles/)
# Kernel statistics are those of the TIMED dispatches of each profiled bench.py run (tools/summarize_prof.py); the PMC passes run at a batch
# that launches the kernel the bench times (PNG: 64 images = 2 176 (image, band) units: the work-queue kernel).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/profile.sh r04_jpeg 64 -- --steps 30 --warmup 10
bash tools/profile.sh r04_jpeg_photo 64 -- --workload jpeg:photo --steps 30 --warmup 10
bash tools/profile.sh r04_jpeg_444 64 -- --workload jpeg:4:1 --steps 20 --warmup 5
bash tools/profile.sh r04_jpeg_422 64 -- --workload jpeg:4:2 --steps 20 --warmup 5
bash tools/profile.sh r04_jpeg_440 64 -- --workload jpeg:4:3 --steps 20 --warmup 5
bash tools/profile.sh r04_png_random 64 -- --workload png --steps 10 --warmup 3
bash tools/profile.sh r04_png 64 -- --workload png:heuristic --steps 10 --warmup 3
PER_STEP=4 bash tools/profile.sh r04_convert_rgba16_rgbaf32 2 -- --workload convert:rgba16:rgbaf32 --batch 256 --steps 6 --warmup 2
PER_STEP=3 bash tools/profile.sh r04_convert_rgbaf32_rgba8 2 -- --workload convert:rgbaf32:rgba8 --batch 256 --steps 6 --warmup 2
PER_STEP=2 bash tools/profile.sh r04_convert_rgba8_rgba16 2 -- --workload convert:rgba8:rgba16 --batch 256 --steps 6 --warmup 2
# the mixed step at config 5's size on one GPU, the file-level feeders: kernel trace only
trace() {   # trace <tag> <timed dispatches per kernel> -- <command...>
  local tag=$1 steps=$2; shift 3
  mkdir -p $R/gpurun_out/summary_$tag
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$tag && mkdir -p /tmp/tr_$tag &&
   timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/tr_$tag/trace -o t -- "$@" > $R/gpurun_out/summary_$tag/run.txt 2>&1;
   python $R/tools/summarize_prof.py /tmp/tr_$tag $R/gpurun_out/summary_$tag 0 $steps | head -12; rm -rf /tmp/tr_$tag)
}
trace r04_mixed_8192 10 -- python $R/bench.py --workload mixed --total-images 8192 --steps 10 --warmup 3 --no-cpu --no-traffic --no-also
trace r04_jpeg_files 0 -- python $R/tools/e2e_bench.py --batch 1024 --paths c --reps 4
trace r04_progressive_files 0 -- python $R/tools/e2e_bench.py --batch 1024 --paths c --reps 2 --progressive
trace r04_png_files 0 -- python $R/tools/e2e_png_bench.py --batch 256
# PNG read traffic at the bench's own batch (512): the counters behind the aligned loads
QUEUES=1 bash tools/png_reads.sh base 512 "png png:heuristic" > gpurun_out/r04_png_reads_512.txt 2>&1
GAMUT_HIP_PNG_ALIGNED=0 QUEUES=1 bash tools/png_reads.sh base 512 "png:heuristic" >> gpurun_out/r04_png_reads_512.txt 2>&1
# what a caller with files sees; the same with two host threads (a rank's share on an 8-rank node)
python tools/files_bench.py > gpurun_out/r04_files_bench.jsonl 2>/dev/null
GAMUT_HIP_HOST_THREADS=2 python tools/files_bench.py > gpurun_out/r04_files_bench_2threads.jsonl 2>/dev/null
python tools/e2e_mixed_bench.py --batch 768 > gpurun_out/r04_mixed_e2e.txt 2>&1; python tools/e2e_mixed_bench.py --batch 3072 >> gpurun_out/r04_mixed_e2e.txt 2>&1
bash tools/pmc_valu.sh > gpurun_out/summary_r04_pmc_valu.txt 2>&1
# progressive: per scan when it ran, how long, how much of it waiting (GAMUT_HIP_TRACE); the two microbenchmarks behind the round's decisions
for b in 256 1024 4096; do GAMUT_HIP_TRACE=1 timeout 300 python tools/e2e_bench.py --batch $b --paths c --reps 2 --progressive 2>&1 | grep -v "amdgpu" | tail -12; done > gpurun_out/r04_progressive_final.txt 2>&1
(hipcc --offload-arch=gfx950 -O2 tools/microbench/chain_latency.hip -o /tmp/cl 2>/dev/null && timeout 60 /tmp/cl) > gpurun_out/r04_chain_latency.txt 2>&1
(hipcc --offload-arch=gfx950 -O2 tools/microbench/h2d_streams.hip -o /tmp/h2d 2>/dev/null && timeout 120 /tmp/h2d) > gpurun_out/r04_h2d_streams.txt 2>&1
(time python bench.py) > gpurun_out/r04_bench_default.log 2>&1
