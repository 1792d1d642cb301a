#!/bin/bash
# k_modp_twin_exp_buckets_pair with and without its memory operations (TWIN_DIAG_NOMEM=1 build, wrong results): the shortest launch
# of lone dealer boxes under rocprofv3 --kernel-trace --stats.   tools/build_lib_variant.sh nomem "-DTWIN_DIAG_NOMEM=1" first.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in shipped nomem; do
  if [ $v = nomem ]; then export MPVSS_HIP_LIB=ab_libs/libmpvss_hip_nomem.so; fi
  rm -rf gpurun_out/twin_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/twin_$v -- python3 tools/bench_dealer.py 65536 4 1 > gpurun_out/twin_$v.log 2>&1
  echo "$v: $(grep -h k_modp_twin_exp_buckets_pair $(find gpurun_out/twin_$v -name '*kernel_stats.csv') | head -1)"
  rm -rf gpurun_out/twin_$v
done
