cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/jt -o t -- python $R/tools/e2e_bench.py --batch 1024 --paths c --reps 4 > /tmp/jt.log 2>&1
grep "files ->" /tmp/jt.log
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/jt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gamut' in r['Name']:
            print(r['Name'][:80], r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
