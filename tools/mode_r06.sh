#!/bin/bash
# Round 6: what selects the slow mode of the PNG heuristic case?  
#   (1) tools/png_mode_probe.py in PROCS separate processes: does the mode flip inside a process when only the buffers' placement changes?
#   (2) per-process counters next to the same process's own launch time: address translation (UTCL1), L2 hit / miss and fabric requests,
#       read latency.  One counter set per process (the mode is a property of the process, so every pass carries its own timing).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
OUT=$R/gpurun_out/${OUT:-r06_png_mode_probe.txt}
: > $OUT
for i in $(seq 1 ${PROCS:-4}); do
  echo "== process $i  $(date +%T)" >> $OUT
  timeout 400 python tools/png_mode_probe.py --trials ${TRIALS:-2} ${PROBE_ARGS} >> $OUT 2>&1
done
cd /tmp; export TMPDIR=/tmp
PM=$R/gpurun_out/${PMOUT:-r06_png_mode_counters.txt}
: > $PM
for rep in $(seq 1 ${REPS:-3}); do
for ctrs in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"; do
  D=/tmp/pmc_mode; rm -rf $D
  echo "== rep $rep: $ctrs" >> $PM
  timeout 400 rocprofv3 --output-format csv --kernel-include-regex gamut --pmc $ctrs -d $D -o t -- python $R/tools/png_mode_probe.py --only initial ${PROBE_ARGS} > /tmp/pmc_mode.log 2>&1
  grep -h "initial#" /tmp/pmc_mode.log >> $PM
  python - $D >> $PM <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_png_defilter" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
if not acc: print("   no counter rows:", open("/tmp/pmc_mode.log").read()[-300:].replace("\n", " | "))
for c, (s, n) in sorted(acc.items()): print(f"   {c:44s} {s / n:18.1f} per launch (n={n})")
PY
done
done
