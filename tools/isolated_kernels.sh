#!/bin/bash
# Per-kernel durations with every kernel ALONE on the chip: one hardware queue (GPU_MAX_HW_QUEUES=1 serialises the streams), boxes
# verified one at a time, rocprofv3 --kernel-trace --stats.   tools/isolated_kernels.sh OUTDIR NAME ENV=VAL ... [-- NAME2 ENV=VAL ...]
# writes gpurun_out/OUTDIR/NAME_kernel_stats.csv (Name, Calls, TotalDurationNs, AverageNs, ...) per variant.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
while [ $# -gt 0 ]; do
  NAME=$1; shift
  ENVS=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  for e in "${ENVS[@]}"; do export "$e"; done
  export GPU_MAX_HW_QUEUES=1 MPVSS_BENCH_DEPTH=1 MPVSS_PIPELINED=1
  rm -rf $OUT/tmp_$NAME
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tmp_$NAME -- python3 bench.py --steps 3 --warmup 1 --lone-boxes 0 --cpu-sample 0 \
      --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --steady-steps 0 > $OUT/$NAME.log 2>&1
  cp $(find $OUT/tmp_$NAME -name "*kernel_stats.csv" | head -1) $OUT/${NAME}_kernel_stats.csv
  rm -rf $OUT/tmp_$NAME
  for e in "${ENVS[@]}"; do unset "${e%%=*}"; done
  python3 - $OUT/${NAME}_kernel_stats.csv $NAME <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("==", sys.argv[2])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:44]:44s} calls {int(r["Calls"]):5d} total {float(r["TotalDurationNs"])/1e6:9.1f} ms avg {float(r["AverageNs"])/1e6:8.3f} ms')
P
done
