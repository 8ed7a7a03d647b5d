timeout 300 python -m pytest tests/test_qoi_gpu.py tests/test_oob_gpu.py tests/test_image_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --workload mixed --steps 5 --warmup 1 --no-cpu --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('mixed', r['value'], 'Mpx/s', r['ms_per_step'], 'ms', {k:v['ms'] for k,v in r['config']['per_format'].items()})"
timeout 300 python bench.py --workload mixed --batch 768 --steps 5 --warmup 1 --no-cpu --no-traffic 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); print('mixed 768', r['value'], 'Mpx/s', r['ms_per_step'], 'ms', {k:v['ms'] for k,v in r['config']['per_format'].items()})"
