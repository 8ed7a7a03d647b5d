#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel trace + two PMC passes for the headline kernel.
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o jpeg -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu > $OUT/trace.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o jpeg -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --batch 256 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o jpeg -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --batch 256 > $OUT/pmc_write.log 2>&1
for f in $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); do echo "== $f"; head -3 $f; done
find $OUT -type f | head -20
python $REPO/tools/summarize_prof.py $OUT $REPO/gpurun_out/summary_$TAG
cp $OUT/bench.json $REPO/gpurun_out/summary_$TAG/bench.json
tail -3 $OUT/trace.log
rm -rf $OUT
cat $REPO/gpurun_out/summary_$TAG/kernel_stats.csv | head -12
