#!/bin/bash
# A/B of one GPU's slice of BASELINE config C5 (n = 131072, t = 1024) through bench.py's `configs` leg, variants interleaved:
#   tools/ab_c5.sh OUTFILE [-r REPS] [-k BOXES] -- NAME ENV=VAL ... -- NAME2 ENV=VAL ...
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; shift
mkdir -p "$(dirname "$OUT")"
REPS=1; BOXES=20
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case "$1" in -r) REPS=$2; shift 2;; -k) BOXES=$2; shift 2;; *) echo "bad option $1"; exit 2;; esac
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
    env ${ENVSTR[$v]} MPVSS_BENCH_CONFIGS=c5_slice MPVSS_BENCH_DETAIL=/tmp/ab_c5_detail.json python3 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 \
        --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes $((BOXES * 8)) --lone-boxes 0 --steady-steps 0 --drop-in-threads 0 >/dev/null 2>gpurun_out/ab_c5_err.txt
    python3 -c "
import json
d = json.load(open('/tmp/ab_c5_detail.json'))
c = d['configs']['c5_slice']
print('$NAME', round(c['value']), round(c['ms_per_box'], 1), 'frac', round(c['compute']['frac'], 3), {k: round(v, 1) for k, v in c['host_ms_per_box'].items()},
      'fallbacks', d['compute']['fd_fallbacks'], d.get('secondary_error'))" | tee -a "$OUT"
  done
done
