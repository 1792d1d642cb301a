#!/bin/bash
# A/B of the registered-keys leg of bench.py (headline shape, K = 20), variants interleaved:
#   tools/ab_keyset.sh OUTFILE [-r REPS] -- NAME ENV=VAL ... -- NAME2 ENV=VAL ...
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; shift
mkdir -p "$(dirname "$OUT")"
REPS=1
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case "$1" in -r) REPS=$2; shift 2;; *) echo "bad option $1"; exit 2;; esac
done
NAMES=(); ENVSTR=()
while [ $# -gt 0 ]; do
  shift
  NAMES+=("$1"); shift
  E=""
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do E="$E $1"; shift; done
  ENVSTR+=("$E")
done
for rep in $(seq $REPS); do
  for v in "${!NAMES[@]}"; do
    NAME=${NAMES[$v]}
    # shellcheck disable=SC2086
    env ${ENVSTR[$v]} MPVSS_BENCH_DETAIL=/tmp/ab_ks_detail.json python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 1 \
        --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 0 --steady-steps 0 --drop-in-threads 0 >/dev/null 2>gpurun_out/ab_ks_err.txt
    python3 -c "
import json
d = json.load(open('/tmp/ab_ks_detail.json'))
k = d['registered_keys']
print('$NAME', 'headline', round(d['value']), 'registered', round(k['value'] or 0), round(k.get('ms_per_step') or 0, 2), 'GB', round((k.get('table_bytes') or 0) / 1e9, 1), d.get('secondary_error'))" | tee -a "$OUT"
  done
done
