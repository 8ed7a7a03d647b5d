# rocprofv3 counters of the QOI launch of tools/qoi_bench.py (2730 streams): instruction mix, busy / wait cycles.  GAMUT_HIP_LIB picks the library.
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/qpmc_tmp; rm -rf $OUT; mkdir -p $OUT
QOI_BENCH_B=2730 timeout 300 rocprofv3 --output-format csv --pmc $set -d $OUT -o t -- python $GRAFT_REPO_ROOT/tools/qoi_bench.py > $OUT/log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "qoi" not in r["Kernel_Name"]: continue
        a = acc[r["Kernel_Name"][:60]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (s, n) in sorted(d.items()): print(f"   {c:28s} {s/n:18.1f}  (n={n})")
PY
rm -rf $OUT
done
