cd $GRAFT_REPO_ROOT
python -m pytest tests/test_jpeg_gpu.py -m gpu -x -q 2>&1 | tail -12
for ho in dense tokens; do
GAMUT_HIP_JPEG_HANDOFF=$ho python tools/e2e_bench.py --batch 1024 --paths c --reps 6 2>&1 | grep "files ->"
GAMUT_HIP_JPEG_HANDOFF=$ho python tools/e2e_bench.py --batch 256 --paths c --reps 6 2>&1 | grep "files ->"
done
