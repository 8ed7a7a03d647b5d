#!/bin/bash
# bash tools/pmc.sh "<COUNTER1 COUNTER2 ...>" -- <bench.py args...>   (one rocprofv3 pass; prints per-kernel averages)
set -u
CTR=$1; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_tmp; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --output-format csv --kernel-include-regex gamut --pmc $CTR -d $OUT -o t -- python $REPO/bench.py "$@" --no-cpu --no-also > $OUT/log 2>&1
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
tail -2 $OUT/log | cut -c1-300
rm -rf $OUT
