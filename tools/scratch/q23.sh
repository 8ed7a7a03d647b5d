timeout 300 python -m pytest tests/test_qoi_gpu.py -m gpu -x -q 2>&1 | tail -5
GAMUT_HIP_QOI_PIPE=0 timeout 300 python -m pytest tests/test_qoi_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/qoi_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
for e in 0 1; do
GAMUT_HIP_QOI_PIPE=$e timeout 300 python bench.py --workload mixed --steps 5 --warmup 1 --no-cpu --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('mixed pipe=$e', r['value'], 'Mpx/s', r['ms_per_step'], 'ms', {k:v['ms'] for k,v in r['config']['per_format'].items()})"
done
