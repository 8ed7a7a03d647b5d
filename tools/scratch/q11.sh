timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for WL in "png" "png:random" "mixed"; do
timeout 300 python bench.py --workload $WL --steps 10 --warmup 2 --no-cpu --no-traffic 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['config']['workload'][:70], r['value'], 'Mpx/s', r['ms_per_step'], 'ms', r['roofline']['achieved'], 'GB/s', r['config'].get('per_format'))
"
done
