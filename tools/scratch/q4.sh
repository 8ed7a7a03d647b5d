GAMUT_HIP_PNG_QUEUE=1 timeout 900 python -m pytest tests/test_png_gpu.py tests/test_oob_gpu.py -m gpu -x -q 2>&1 | tail -4
bash tools/scratch/q3.sh
