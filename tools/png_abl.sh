#!/bin/bash
# bash tools/png_abl.sh <name>:<flags> ...   -- build libgamut_hip variants whose png.hip is compiled with extra flags
# (tuning experiments, e.g. nt0:-DPNG_NT_STORES=0; the ablation switches that located the unaligned write-back cost --
# no loads / no stores / aligned stores / no cross-band wait, DESIGN.md 4.3 -- were temporary and are gone) into
# gamut_amd/lib/var/; run on the box with:  GAMUT_HIP_LIB=gamut_amd/lib/var/libgamut_hip_<name>.so python bench.py --workload png
set -e
cd "$(dirname "$0")/../gamut_amd/csrc"
mkdir -p ../lib/var build/var
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c png.hip -o build/var/png_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libgamut_hip_$name.so $(ls build/*.o | grep -v /png.o) build/var/png_$name.o -lz ) &
done
wait
ls -la ../lib/var
