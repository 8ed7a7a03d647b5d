#!/bin/bash
# rocprofv3 --kernel-trace --stats of the file-level feeders (run on the GPU box): baseline / progressive JPEG batches, QOI streams.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
prof() {   # tag, command...
  tag=$1; shift
  rm -rf /tmp/fp_$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_$tag -o t -- "$@" > /tmp/fp_$tag.log 2>&1
  echo "== $tag: $*"
  python - /tmp/fp_$tag <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gamut" in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows: print(f"   {r['Name'][:96]:96s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e6:9.3f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
}
prof jpeg_baseline_1024 python $REPO/tools/e2e_bench.py --batch 1024 --reps 2
GAMUT_HIP_JPEG_PROGRESSIVE=device prof jpeg_progressive_1024 python $REPO/tools/e2e_bench.py --progressive --batch 1024 --reps 1
QOI_BENCH_B=256,341 prof qoi_256_341 python $REPO/tools/qoi_bench.py
