for e in 0 1; do echo "== pipe=$e"; GAMUT_HIP_QOI_PIPE=$e QOI_BENCH_B=64,256,341,512,700 timeout 300 python tools/qoi_bench.py 2>&1 | grep -v amdgpu.ids | head -7; done
