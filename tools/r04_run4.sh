cd $GRAFT_REPO_ROOT
for al in 1 0; do
echo "=== aligned=$al png random 512"
GAMUT_HIP_PNG_ALIGNED=$al GAMUT_BENCH_NOCHECK=1 bash tools/pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" -- --workload png --steps 3 --warmup 1 --no-traffic
GAMUT_HIP_PNG_ALIGNED=$al GAMUT_BENCH_NOCHECK=1 bash tools/pmc.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" -- --workload png --steps 3 --warmup 1 --no-traffic
done 2>&1 | grep -v "^$" | tee gpurun_out/r04_png_sq.txt
