cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
WLS="jpeg" GEOMS="4096x2160x256" 
for spec in "jpeg 4096 2160 256" "jpeg:3 1366 768 2048" "jpeg:4:1 2048 2048 512" "jpeg:4:2 2000 2000 512" "jpeg:4:0 2048 2048 512" "png:heuristic 2000 2000 512" "png:random 1080 1920 1024" "png:random:3:4 1088 1920 1024" "png:heuristic 1366 768 2048"; do
  set -- $spec
  timeout 200 python bench.py --workload $1 --width $2 --height $3 --batch $4 --steps 10 --warmup 3 --no-cpu --no-traffic --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('rep $rep %-22s %5d x %-5d batch %-5d %8.3f ms (min %.3f) frac %.3f' % ('$1', $2, $3, $4, r['roofline']['kernel_ms_avg'], r['roofline']['kernel_ms_min'], r['roofline']['frac']))
"
done; done > gpurun_out/r06_geometry_repeat.txt 2>&1
