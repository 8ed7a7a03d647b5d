#!/bin/bash
# bash tools/variant.sh <file>:<name>:<flags> ...   -- build libgamut_hip variants in which <file>.hip is compiled with extra
# flags (tuning knobs such as png:nt0:-DPNG_NT_STORES=0 or convert:nt1:-DCONVERT_NT=1) into gamut_amd/lib/var/; run on the
# GPU box with  GAMUT_HIP_LIB=gamut_amd/lib/var/libgamut_hip_<name>.so python bench.py ...
set -e
cd "$(dirname "$0")/../gamut_amd/csrc"
make -s -j8
mkdir -p ../lib/var build/var
for spec in "$@"; do
  file=${spec%%:*}; rest=${spec#*:}; name=${rest%%:*}; flags=${rest#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c $file.hip -o build/var/${file}_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/var/libgamut_hip_$name.so $(ls build/*.o | grep -v /$file.o) build/var/${file}_$name.o -lz -lpthread ) &
done
wait
ls ../lib/var
