#!/bin/bash
# bash tools/feeder_pmc.sh  (GPU box): instruction / cycle counters of the file-level feeder kernels (one rocprofv3 --pmc pass each set)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
OUT=/tmp/fpmc; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --output-format csv --pmc $set -d $OUT -o t -- python $REPO/tools/e2e_bench.py --batch ${1:-1024} --reps 1 ${2:-} > $OUT/log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gamut" not in r["Kernel_Name"]: continue
        a = acc[r["Kernel_Name"][:70]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (s, n) in sorted(d.items()): print(f"   {c:28s} {s/n:18.1f}  (n={n})")
PY
done
