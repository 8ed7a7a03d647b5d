#!/bin/bash
# Every bench.py workload once (same box, same build) -> gpurun_out/bench_matrix.jsonl; copy into profiles/ to commit.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/bench_matrix.jsonl; mkdir -p $REPO/gpurun_out; : > $OUT
run() { timeout 400 python $REPO/bench.py --no-cpu "$@" 2>/dev/null | tail -1 >> $OUT; }
run --workload jpeg
for w in jpeg:3 jpeg:1; do run --workload $w; done
for w in jpeg:4:1 jpeg:3:1 jpeg:4:2 jpeg:4:3 jpeg:3:3 jpeg:1:0; do run --workload $w --batch 512; done
for w in png:heuristic png:random png:4 png:heuristic:3 png:random:3 png:heuristic:1 png:heuristic:2 png:heuristic:3:4; do run --workload $w --steps 10; done
for p in rgba16:rgbaf32 rgbaf32:rgba16 rgba8:rgbaf32 rgbaf32:rgba8 rgba8:rgba16 rgba16:rgba8 rgb8:rgba8 rgba8:l8 l16:rgbaf32 rgbap16:rgbaf32; do run --workload convert:$p --steps 10; done
run --workload mixed --steps 5 --warmup 1
for b in 1 8 64; do run --workload png:random --batch $b --steps 10; done
python - "$OUT" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    print(f"{r['metric'][17:-1] if r['metric'].startswith('Mpixels/sec (') else 'jpeg (headline)':28s} {r['value']:12.1f} Mpx/s  {r['roofline']['achieved']:8.1f} GB/s  {100 * r['roofline']['frac']:5.1f} %   {r['config']['workload']}")
PY
