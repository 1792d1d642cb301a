#!/bin/bash
# rocprofv3 kernel trace of bench.py's C5-slice leg (n = 131072, t = 1024, 12 boxes) under the given environment; summary by tools/trace_occupancy.py
#   tools/trace_c5.sh NAME ENV=VAL ...
set -u
cd "$(dirname "$0")/.."
NAME=$1; shift
OUT=gpurun_out/trace_c5_$NAME
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp MPVSS_BENCH_CONFIGS=c5_slice MPVSS_BENCH_DETAIL=/tmp/trace_c5_detail.json
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -- python3 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 --wb-shares 0 --registered-keys 0 \
  --ec-boxes 0 --host-boxes 0 --config-boxes 96 --lone-boxes 0 --steady-steps 0 > $OUT/line.json 2> $OUT/err.txt
CSV=$(find $OUT/raw -name '*kernel_trace.csv' | head -1)
python3 tools/trace_occupancy.py "$CSV" k_modp_fd_step_pair > $OUT/summary.txt 2>&1
python3 -c "
import json
d = json.load(open('/tmp/trace_c5_detail.json'))['configs']['c5_slice']
print('c5_slice', round(d['value']), round(d['ms_per_box'], 1))" >> $OUT/summary.txt
rm -rf $OUT/raw
cat $OUT/summary.txt
