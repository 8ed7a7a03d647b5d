tools/scratch/perm_probe
timeout 900 python -m pytest tests/test_png_gpu.py -m gpu -q 2>&1 | grep -E "^E  .*(differ|assert)|passed|failed" | head -40
