#!/bin/bash
# Round 6's soak of the fuzzers on the GPU box (new seeds): every file-level path against the oracle after this round's changes (PNG write-back by
# memory lines, QOI boundary walk, status words of host-feeder-only JPEG batches).  One summary line per fuzzer -> gpurun_out/r06_soak.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
{
for spec in "fuzz_mixed_gpu.py 40 ${SOAK_SEED:-61}" "fuzz_mixed_gpu.py 25 $((${SOAK_SEED:-61}+1))" "fuzz_png_gpu.py 30 $((${SOAK_SEED:-61}+2))" "fuzz_input_gpu.py 40 $((${SOAK_SEED:-61}+3))" "fuzz_prog_gpu.py 15" "fuzz_prog_gpu.py 15 baseline" "fuzz_qoi_gpu.py 60 $((${SOAK_SEED:-61}+4))" "fuzz_inflate.py 3000 $((${SOAK_SEED:-61}+5))"; do
  set -- $spec
  echo "== $spec  $(date +%T)"
  timeout 900 python tools/$1 ${@:2} 2>&1 | tail -3
done
GAMUT_FUZZ_HEADERS=1 timeout 600 python tools/fuzz_mixed_gpu.py 25 $((${SOAK_SEED:-61}+6)) 2>&1 | tail -2
} > gpurun_out/r06_soak.txt 2>&1
