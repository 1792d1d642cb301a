#!/bin/bash
# Builds tools/mfma_mont/ubench_<name> for a list of "name:flags" variants (bn_pair.h compile-time switches) and prints each
# kernel's registers / spills:   tools/mfma_mont/build_variants.sh base: pf2:-DMM_PREFETCH=2 ...
# Every variant is built as the product builds the a2 kernel: two-wave workgroups, registers for two waves per SIMD.
set -u
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  $HIPCC -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 -DUB_WAVES=2 -DUB_WPE=2 $flags \
      -Rpass-analysis=kernel-resource-usage ubench.hip -o ubench_$name 2> /tmp/ubench_$name.log || { tail -5 /tmp/ubench_$name.log; continue; }
  python3 - "$name" /tmp/ubench_$name.log <<'P'
import re, sys
name, txt = sys.argv[1], open(sys.argv[2]).read()
for k in ("k_chain_pairILb0", "k_chain_pairILb1"):
    i = txt.find(k)
    blk = txt[i:i + 3000]
    g = lambda pat: re.search(pat + r"[^:\n]*: (\d+)", blk).group(1)
    print(name.ljust(10), "squaring" if "Lb0" in k else "product ", "VGPRs", g(" VGPRs"), "AGPRs", g("AGPRs"), "scratch", g("ScratchSize"),
          "occupancy", g("Occupancy"), "LDS", g("LDS Size"))
P
done
