#!/bin/bash
# The inflate kernel's profile set, on the GPU box:  bash tools/inflate_profile.sh   -> gpurun_out/summary_r03_inflate/ (copy into profiles/)
# needs gamut_amd/lib/var/libgamut_hip_prof.so (bash tools/variant.sh inflate:prof:-DINFLATE_PROFILE=1) for the phase counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/summary_r03_inflate; mkdir -p $O
python tools/inflate_bench.py --reps 3 > $O/inflate_bench.txt 2>&1
GAMUT_HIP_LIB=gamut_amd/lib/var/libgamut_hip_prof.so python tools/inflate_bench.py --reps 1 2>&1 | grep -v "repetitions" > $O/inflate_phases.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/inf_tr -o t -- python $R/tools/inflate_bench.py --reps 2 > /dev/null 2>&1; cp $(find /tmp/inf_tr -name "*kernel_stats.csv" | head -1) $R/$O/kernel_stats.csv; rm -rf /tmp/inf_tr)
python tools/e2e_png_bench.py --batch 256 > $O/png_e2e.txt 2>&1
python tools/e2e_png_bench.py --batch 256 --content smooth >> $O/png_e2e.txt 2>&1
python tools/fuzz_inflate.py 20000 21 2>&1 | tail -2 > $O/fuzz_inflate.txt
