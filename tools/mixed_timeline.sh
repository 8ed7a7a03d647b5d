#!/bin/bash
# Which kernel of config 5's mixed step runs when: rocprofv3 --kernel-trace of three steps of `bench.py --workload mixed --total-images 8192`,
# the last step's kernels with start / end relative to the step's first kernel -> gpurun_out/r06_mixed_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/mixed_tl; rm -rf $OUT; mkdir -p $OUT
GAMUT_BENCH_NOCHECK=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --workload mixed --total-images 8192 --steps 3 --warmup 1 --no-cpu --no-traffic --no-also $MIXED_ARGS > $OUT/log 2>&1
python - "$OUT" <<'PY' > $R/gpurun_out/${MIXED_TL_NAME:-r06_mixed_timeline}.txt
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gamut" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), ("png_defilter" if "png" in r["Kernel_Name"] else "qoi_decode" if "qoi" in r["Kernel_Name"] else "jpeg" if "jpeg" in r["Kernel_Name"] else r["Kernel_Name"][:40]), r.get("Stream_Id", r.get("Queue_Id", "?")), r.get("LDS_Block_Size", "?")))
rows.sort()
# the last step: the kernels after the last gap of more than 2 ms... take the last 12 kernels
last = rows[-10:]
t0 = last[0][0]
for s, e, n, q, lds in last:
    print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms  ({(e - s) / 1e6:7.3f})  queue {q}  lds {lds}  {n}")
PY
rm -rf $OUT
