#!/bin/bash
# A copy of the engine library with other compile-time switches of the pair-layout kernels, for tools/ab_bench.sh:
#   tools/build_lib_variant.sh NAME "-DPAIR_PREFETCH=0 -DMM_SGB=0"   ->  ab_libs/libmpvss_hip_NAME.so   (MPVSS_HIP_LIB=ab_libs/...)
# Only modp_pair_kernels.o is rebuilt with the flags; the default library is restored afterwards.
set -eu
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
mkdir -p ab_libs
make -s -C mpvss_rs_amd/csrc -B modp_pair_kernels.o PAIR_EXTRA="$FLAGS"
make -s -C mpvss_rs_amd/csrc
cp mpvss_rs_amd/libmpvss_hip.so ab_libs/libmpvss_hip_$NAME.so
make -s -C mpvss_rs_amd/csrc -B modp_pair_kernels.o
make -s -C mpvss_rs_amd/csrc
echo "ab_libs/libmpvss_hip_$NAME.so"
