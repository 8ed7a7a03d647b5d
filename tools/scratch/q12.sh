timeout 600 python -m pytest tests/test_jpeg_gpu.py -m gpu -x -q -k "entropy or progressive" 2>&1 | tail -30
