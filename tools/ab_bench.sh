#!/bin/bash
# One A/B line of the headline pipeline per variant (replaces the per-experiment scripts of earlier rounds):
#   tools/ab_bench.sh OUTFILE [-k STEPS] [-r REPS] [-s STEADY_STEPS] -- NAME1 ENV=VAL ... -- NAME2 ENV=VAL ... 
# Every variant runs `bench.py --gpus 1 --steps STEPS --warmup 5` with the secondary figures switched off, REPS times, under
# the given environment, and appends "NAME value ms_per_step a2_alone_ms kernel_ms_sums" to gpurun_out/OUTFILE.
# Variant libraries (e.g. a kernel built with other -D flags): make -C mpvss_rs_amd/csrc PAIR_EXTRA=... and pass
# MPVSS_HIP_LIB=/path/to/copy.so as one of the ENV=VAL words.
# Examples of round 3 (profiles/r03_pair_ab.txt, r03_tail_ab.txt):
#   tools/ab_bench.sh r03_pair/ab.txt -- pair MPVSS_PAIR=1 -- all_pair MPVSS_PAIR=15 -- quad MPVSS_PAIR=0
#   tools/ab_bench.sh r03_tail/ab.txt -r 3 -- plain MPVSS_TAIL_PARTS=1 -- last1x4 MPVSS_TAIL_PARTS=4 MPVSS_TAIL_BOXES=1
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/$1; shift
mkdir -p "$(dirname "$OUT")"
STEPS=20; REPS=1; STEADY=0
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case "$1" in -k) STEPS=$2; shift 2;; -r) REPS=$2; shift 2;; -s) STEADY=$2; shift 2;; *) echo "bad option $1"; exit 2;; esac
done
# variants are collected first and run INTERLEAVED (rep 1 of every variant, then rep 2, ...): the boxes of the pool drift by a few
# per cent over minutes, which a block of runs per variant would read as a difference between variants
NAMES=(); ENVSTR=()
while [ $# -gt 0 ]; do
  shift                      # the "--"
  NAMES+=("$1"); shift
  E=""
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do E="$E $1"; shift; done
  ENVSTR+=("$E")
done
for rep in $(seq $REPS); do
  for v in "${!NAMES[@]}"; do
    NAME=${NAMES[$v]}
    # shellcheck disable=SC2086
    env ${ENVSTR[$v]} MPVSS_BENCH_DETAIL=/tmp/ab_detail.json python3 bench.py --gpus 1 --steps $STEPS --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 \
        --host-boxes 0 --config-boxes 0 --lone-boxes 1 --steady-steps $STEADY --drop-in-threads 0 >/dev/null 2>/dev/null
    python3 -c "
import json
d = json.load(open('/tmp/ab_detail.json'))
print('$NAME', round(d['value']), round(d['ms_per_step'], 2), round(d['roofline']['kernel_ms'], 1), 'steady', round(d['value_steady_state'] or 0),
      {k: round(v) for k, v in d['compute']['kernel_ms_sums'].items() if k != 'note'},
      {k: round(v, 1) for k, v in d['host']['per_box_ms'].items()})" | tee -a "$OUT"
  done
done
