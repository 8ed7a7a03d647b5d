timeout 900 python bench.py --workload mixed --total-images 8192 --steps 3 --warmup 1 --no-cpu --no-traffic 2>&1 | tail -2 | cut -c1-1500
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 256 --no-cpu --no-traffic 2>&1 | tail -1 | cut -c1-600
