cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
[ -n "$SKIPTEST" ] || timeout 1200 python -m pytest tests/test_jpeg_gpu.py tests/test_oob_gpu.py -x -q 2>&1 < /dev/null | tail -5
for rep in 1 2; do
for v in ${VARIANTS:-prev base}; do
  if [ $v = base ]; then L=gamut_amd/lib/libgamut_hip.so; else L=gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for wl in ${WLS:-jpeg:3 jpeg:1}; do
    GAMUT_HIP_LIB=$L timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu --no-also --no-traffic 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('$v', '$wl', r['roofline']['kernel_ms_avg'], 'ms', r['roofline']['frac'], r['config']['parity_check'][:50])
    elif 'PARITY' in l or 'Error' in l: print(l.strip()[:200])
"
  done
done
done 2>&1 | tee gpurun_out/${OUT:-r05_jpeg_ab.txt}
