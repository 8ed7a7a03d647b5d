#!/bin/bash
# AddressSanitizer build of the library's HOST code (parsers, feeders) + a mutation fuzz run (see tools/fuzz_host.py).
# (ASan + UBSan.)  Found so far: a heap overflow in HuffTable::build() on DHT segments whose length counts over-subscribe the
# code space; a shift by -1 in the slow path of BitReader::decode (peek(17) with 16 bits buffered; result was unused).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/asan; mkdir -p $OUT
make -s -C $ROOT/gamut_amd/csrc -j8                      # the ordinary objects (kernels) are linked in as they are
cd $ROOT/gamut_amd/csrc
for f in jpeg_host png_host runtime qoi image_host stream_host batch_host; do    # host code of the parsers, instrumented (-Xarch_host: not the device side)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer -c $f.hip -o $OUT/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $OUT/libgamut_hip_asan.so $OUT/jpeg_host.o $OUT/png_host.o $OUT/runtime.o $OUT/qoi.o $OUT/image_host.o $OUT/stream_host.o $OUT/batch_host.o \
    build/jpeg.o build/png.o build/convert.o build/inflate.o build/comm.o build/flip.o -lz -lpthread -ldl
ASAN_RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $ROOT
LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 GAMUT_HIP_LIB=$OUT/libgamut_hip_asan.so python tools/fuzz_host.py "${1:-3000}"
