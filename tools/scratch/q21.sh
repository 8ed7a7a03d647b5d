timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 256 --no-cpu --no-traffic 2>&1 | tail -40 | cut -c1-400
