#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes for one bench workload.
#   bash tools/profile.sh <tag> <pmc-batch> -- <bench.py args...>
# Condensed summaries land in gpurun_out/summary_<tag>/ (copy what you want judged into profiles/).
# The two counter passes skip bench.py's parity check (GAMUT_BENCH_NOCHECK: the counters are the kernel's whatever the host compares afterwards; round 5's
# FETCH_SIZE pass of the headline ran into its time limit inside the check and left no rows); the timed run and the kernel-trace run keep it.
set -u
TAG=$1; PB=$2; shift 3
# timed steps of the profiled run (bench.py's default is 50) and launches per step (PER_STEP, default 1): summarize_prof.py reports the
# kernel statistics over the timed dispatches only
STEPS=50; prev=""; for a in "$@"; do [ "$prev" = "--steps" ] && STEPS=$a; prev=$a; done
PER_STEP=${PER_STEP:-1}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $REPO/bench.py "$@" --no-also > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o t -- python $REPO/bench.py "$@" --no-cpu --no-also --no-traffic > $OUT/trace.log 2>&1
GAMUT_BENCH_NOCHECK=1 timeout 400 rocprofv3 --output-format csv --kernel-include-regex gamut --pmc FETCH_SIZE -d $OUT/pmc_fetch -o t -- python $REPO/bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-also --no-traffic --batch $PB > $OUT/pmc_fetch.log 2>&1
GAMUT_BENCH_NOCHECK=1 timeout 400 rocprofv3 --output-format csv --kernel-include-regex gamut --pmc WRITE_SIZE -d $OUT/pmc_write -o t -- python $REPO/bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-also --no-traffic --batch $PB > $OUT/pmc_write.log 2>&1
grep "^{" $OUT/trace.log | tail -1 > $OUT/bench_profiled.json        # the line of the PROFILED run: its HIP-event average and rocprofv3's come from one process
python $REPO/tools/summarize_prof.py $OUT $REPO/gpurun_out/summary_$TAG $PB $STEPS $PER_STEP | head -40
cp $OUT/bench.json $REPO/gpurun_out/summary_$TAG/bench.json
cp $OUT/bench_profiled.json $REPO/gpurun_out/summary_$TAG/bench_profiled.json
rm -rf $OUT
grep gamut $REPO/gpurun_out/summary_$TAG/kernel_stats.csv | cut -c1-200
