#!/bin/bash
# Round 6's profile set, on the GPU box:  bash tools/profile_r06.sh [a|b]   -> gpurun_out/summary_r06_*/ and gpurun_out/r06_*  (copied into profiles/)
# Kernel statistics = the TIMED dispatches of each profiled bench.py run (tools/summarize_prof.py); counter passes at the TIMED batch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
if [ "${1:-a}" = a ]; then
bash tools/profile.sh r06_jpeg 1024 -- --steps 30 --warmup 10
bash tools/profile.sh r06_jpeg_rgb8 1024 -- --workload jpeg:3 --steps 20 --warmup 5
bash tools/profile.sh r06_png_random 512 -- --workload png --steps 10 --warmup 3
bash tools/profile.sh r06_png 512 -- --workload png:heuristic --steps 10 --warmup 3
PER_STEP=4 bash tools/profile.sh r06_convert_rgba16_rgbaf32 2 -- --workload convert:rgba16:rgbaf32 --batch 256 --steps 6 --warmup 2
PER_STEP=3 bash tools/profile.sh r06_convert_rgbaf32_rgba8 2 -- --workload convert:rgbaf32:rgba8 --batch 256 --steps 6 --warmup 2
PER_STEP=2 bash tools/profile.sh r06_convert_rgba8_rgba16 2 -- --workload convert:rgba8:rgba16 --batch 256 --steps 6 --warmup 2
else
PER_STEP=4 bash tools/profile.sh r06_convert_rgbaf32_rgba16 2 -- --workload convert:rgbaf32:rgba16 --batch 256 --steps 6 --warmup 2
PER_STEP=3 bash tools/profile.sh r06_convert_rgba8_rgbaf32 2 -- --workload convert:rgba8:rgbaf32 --batch 256 --steps 6 --warmup 2
PER_STEP=2 bash tools/profile.sh r06_convert_rgba16_rgba8 2 -- --workload convert:rgba16:rgba8 --batch 256 --steps 6 --warmup 2
trace() {   # trace <tag> <timed dispatches per kernel> -- <command...>
  local tag=$1 steps=$2; shift 3
  mkdir -p $R/gpurun_out/summary_$tag
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$tag && mkdir -p /tmp/tr_$tag &&
   timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/tr_$tag/trace -o t -- "$@" > $R/gpurun_out/summary_$tag/run.txt 2>&1;
   python $R/tools/summarize_prof.py /tmp/tr_$tag $R/gpurun_out/summary_$tag 0 $steps | head -12; rm -rf /tmp/tr_$tag)
}
trace r06_mixed_8192 10 -- python $R/bench.py --workload mixed --total-images 8192 --steps 10 --warmup 3 --no-cpu --no-traffic --no-also
python tools/files_bench.py > gpurun_out/r06_files_bench.jsonl 2>/dev/null
(time python bench.py) > gpurun_out/r06_bench_default.log 2>&1
GAMUT_BENCH_BACKEND=gloo python bench.py --gpus 8 --batch 16 --steps 3 --warmup 1 > gpurun_out/r06_bench_8ranks_one_device.json 2> gpurun_out/r06_bench_8ranks_one_device.err
fi
