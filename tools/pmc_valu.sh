for spec in "jpeg|--steps 2 --warmup 1 --batch 64" "png|--workload png:heuristic --steps 2 --warmup 1 --batch 64" "png_paeth|--workload png:4 --steps 2 --warmup 1 --batch 64" "png_rgb8_rgba8|--workload png:heuristic:3:4 --steps 2 --warmup 1 --batch 64" "convert_rgba16_rgbaf32|--workload convert:rgba16:rgbaf32 --steps 2 --warmup 1 --batch 2"; do
  tag=${spec%%|*}; args=${spec#*|}
  echo "== $tag: bench.py $args"
  bash tools/pmc.sh "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS" -- $args 2>&1 | grep -v "rocprofv3\]"
  bash tools/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" -- $args 2>&1 | grep -v "rocprofv3\]" | grep -v "^void"
  timeout 120 python bench.py $args --no-cpu --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('   kernel_ms_avg', r['roofline']['kernel_ms_avg'], ' images', r['config']['images_per_gpu_per_step'])
"
done
