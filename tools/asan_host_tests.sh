#!/bin/bash
# Host side of the library (C ABI, scalar field, wire format, SHA, pipeline bookkeeping) under AddressSanitizer and
# UndefinedBehaviorSanitizer: the host halves of the translation units are rebuilt with -fsanitize (device code is
# untouched: GPU sanitizers are not available on this pool), the CPU test-suite then runs against that library.
set -e
cd "$(dirname "$0")/../mpvss_rs_amd/csrc"
make -s
OUT=${TMPDIR:-/tmp}/mpvss_asan; mkdir -p $OUT
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize"
for f in mpvss_capi sha256 sha512; do /opt/rocm/bin/hipcc $FLAGS -c $f.cpp -o $OUT/$f.o; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -o $OUT/libmpvss_hip.so \
  $OUT/mpvss_capi.o $OUT/sha256.o $OUT/sha512.o modp_kernels.o ec_kernels_secp.o ec_kernels_rist.o verdict_kernels.o
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd ../..
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$RT MPVSS_HIP_LIB=$OUT/libmpvss_hip.so \
  python -m pytest tests -x -q -m "not gpu" "$@"
