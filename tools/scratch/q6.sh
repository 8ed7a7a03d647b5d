for extra in "--serial-formats" ""; do
  timeout 300 python bench.py --workload mixed --steps 5 --warmup 1 --no-cpu $extra 2>/dev/null | tail -1
done
bash tools/profile.sh r02_png_1080p_queue 64 -- --workload png --width 1920 --height 1080 --batch 341 --steps 10 --warmup 2
