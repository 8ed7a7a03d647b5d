cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in ${VARIANTS:-prev base}; do
  if [ $v = base ]; then export GAMUT_HIP_LIB=$PWD/gamut_amd/lib/libgamut_hip.so; else export GAMUT_HIP_LIB=$PWD/gamut_amd/lib/var/libgamut_hip_$v.so; fi
  for wl in ${WLS:-png}; do
  echo "=== $v $wl"
  bash tools/pmc.sh "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" -- --workload $wl --steps 2 --warmup 1 2>&1 | grep -v "rocprofv3\]\|^W2026\|^E2026"
  bash tools/pmc.sh "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" -- --workload $wl --steps 2 --warmup 1 2>&1 | grep -v "rocprofv3\]\|^W2026\|^E2026" | grep -v "^void"
  bash tools/pmc.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" -- --workload $wl --steps 2 --warmup 1 2>&1 | grep -v "rocprofv3\]\|^W2026\|^E2026" | grep -v "^void"
  bash tools/pmc.sh "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM" -- --workload $wl --steps 2 --warmup 1 2>&1 | grep -v "rocprofv3\]\|^W2026\|^E2026" | grep -v "^void"
  done
done 2>&1 | tee gpurun_out/${OUT:-r05_png_sq.txt}
