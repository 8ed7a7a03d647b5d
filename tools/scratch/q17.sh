export GAMUT_HIP_TRACE=1
echo "== tools/e2e_bench.py --progressive (files in host memory -> rgba8 in HBM; A = host feeder + upload, B = gamut_hip_jpeg_entropy_decode_device)"
for B in 64 256 1024 4096; do
GAMUT_HIP_JPEG_PROGRESSIVE=device timeout 900 python tools/e2e_bench.py --progressive --batch $B --reps 2 2>&1 | grep -v "amdgpu.ids" | tail -6
done
unset GAMUT_HIP_TRACE
echo "== tools/e2e_bench.py (baseline files)"
timeout 600 python tools/e2e_bench.py --batch 1024 --reps 2 2>&1 | grep -v "amdgpu.ids" | tail -3
