#!/bin/bash
# On the GPU box: every tools/mfma_mont/ubench_<name> given as argument on the same input (n numbers, S operations each, best of
# 3), one line per variant and kernel, then the exactness check of tools/mfma_mont/run.py on each.
#   tools/mfma_mont/run_variants.sh OUT n S name...
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/$1; N=$2; S=$3; shift 3
mkdir -p "$(dirname "$OUT")"
python3 - "$N" <<'P'
import random, struct, sys
n = int(sys.argv[1]); rng = random.Random(1)
with open('/tmp/in.bin', 'wb') as f:
    for _ in range(n):
        f.write(struct.pack('<72I', *[rng.getrandbits(29) for _ in range(70)] + [rng.getrandbits(18), 0]))
P
for round in 1 2; do
  for v in "$@"; do
    echo "== $v (round $round)" | tee -a "$OUT"
    tools/mfma_mont/ubench_$v /tmp/in.bin /tmp/o $N $S 3 | grep -E "pair|stamps" | tee -a "$OUT"
  done
done
for v in "$@"; do
  echo "== exactness $v" | tee -a "$OUT"
  UBENCH=tools/mfma_mont/ubench_$v python3 tools/mfma_mont/run.py 4096 64 /tmp | grep -E "exact|WRONG" | tee -a "$OUT"
done
