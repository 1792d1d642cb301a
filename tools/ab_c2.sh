#!/bin/bash
# A/B of BASELINE config C2 (n = 4096, t = 64 boxes) through bench.py's `configs` leg:
#   tools/ab_c2.sh OUTFILE BOXES -- NAME ENV=VAL ... -- NAME2 ENV=VAL ...
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; BOXES=$2; shift 2
mkdir -p "$(dirname "$OUT")"
while [ $# -gt 0 ]; do
  shift
  NAME=$1; shift
  ENVS=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
  env "${ENVS[@]}" python3 bench.py --gpus 1 --steps 6 --warmup 2 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 \
      --host-boxes 0 --config-boxes $BOXES --lone-boxes 0 --drop-in-threads 0 2>gpurun_out/ab_c2_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d.get('configs', {})
print('$NAME', 'headline', round(d['value']), 'c2', round(c['c2']['value']), round(c['c2']['ms_per_box'], 3), c['c2']['host_ms_per_box'], d.get('secondary_error'))" | tee -a "$OUT"
done
