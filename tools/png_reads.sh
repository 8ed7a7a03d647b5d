#!/bin/bash
# Where do the PNG de-filter kernels' extra HBM reads come from?  (round 3: 1.086 x the stream for k_png_defilter_queue)
#   bash tools/png_reads.sh "<variants>" "<batches>"      on the GPU box; variants built by tools/variant.sh (base = the product library)
# Per variant x launch shape x batch: kernel time (HIP events) and the L2 / fabric read counters of one rocprofv3 --pmc pass each.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in ${1:-base}; do
  if [ $v = base ]; then L=$R/gamut_amd/lib/libgamut_hip.so; else L=$R/gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for q in ${QUEUES:-1 0}; do
    for b in ${2:-64}; do
      for wl in ${3:-png}; do
      echo "== variant $v queue=$q batch=$b $wl"
      GAMUT_BENCH_NOCHECK=1 GAMUT_HIP_PNG_QUEUE=$q GAMUT_HIP_LIB=$L timeout 300 python $R/bench.py --workload $wl --batch $b --steps 10 --warmup 3 --no-cpu --no-traffic --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('   time', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['achieved'], 'GB/s', r['roofline']['frac'])
"
      for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
        D=/tmp/pmc_png; rm -rf $D
        GAMUT_BENCH_NOCHECK=1 GAMUT_HIP_PNG_QUEUE=$q GAMUT_HIP_LIB=$L timeout 300 rocprofv3 --output-format csv --pmc $ctr -d $D -o t -- python $R/bench.py --workload $wl --batch $b --steps 3 --warmup 1 --no-cpu --no-traffic --no-also > /tmp/pmc_png.log 2>&1
        python - $D $b <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0]); b = int(sys.argv[2])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_png_defilter" not in r["Kernel_Name"]: continue
        a = acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    extra = ""
    if c == "FETCH_SIZE": extra = f"  -> {2 * s / n * 1024 / b / 1e6:.3f} MB read per image (x2 corrected)"
    if c == "WRITE_SIZE": extra = f"  -> {s / n * 1024 / b / 1e6:.3f} MB written per image"
    print(f"   {k:40s} {c:24s} {s / n:16.1f} (n={n}){extra}")
PY
      done
      done
    done
  done
done
