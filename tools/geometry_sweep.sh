#!/bin/bash
# Round 6: the kernels of the path on geometries OTHER than BASELINE.json's -- power-of-two widths (4 KiB / 8 KiB pitches), widths that are no
# multiple of 32 pixels (rows off the 128-byte lines), portrait frames.  One line per (workload, geometry): kernel time, fraction of the HBM peak
# on the algorithmic bytes, Mpx/s.  Cliffs show as a fraction far below the neighbouring geometry's.   bash tools/geometry_sweep.sh [out-file]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/${1:-r06_geometry_sweep.txt}; : > $OUT
run() {  # workload width height batch
  timeout 200 python bench.py --workload $1 --width $2 --height $3 --batch $4 --steps 10 --warmup 3 --no-cpu --no-traffic --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('%-22s %5d x %-5d batch %-5d %8.3f ms  frac %.3f  %9.0f Mpx/s' % ('$1', $2, $3, $4, r['roofline']['kernel_ms_avg'], r['roofline']['frac'], r['value']))
" >> $OUT
}
for wl in ${WLS:-jpeg jpeg:3 jpeg:4:1 jpeg:3:1 jpeg:4:2 jpeg:4:0 png:heuristic png:random png:random:3:4}; do
  for g in ${GEOMS:-"1000 1000 2048" "1024 1024 2048" "1080 1920 1024" "1088 1920 1024" "2000 2000 512" "2048 2048 512" "1366 768 2048" "4096 2160 256"}; do
    set -- $g; run $wl $1 $2 $3
  done
done
cat $OUT
