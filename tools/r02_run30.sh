#!/bin/bash
mkdir -p gpurun_out/r02ac; O=gpurun_out/r02ac
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --ec-boxes 0 --wb-shares 0 --host-boxes 0 --lone-boxes 0"
for rep in a b; do for cfg in "12 6" "10 6" "14 6" "16 6" "12 8" "14 8"; do set -- $cfg
  MPVSS_BENCH_DEPTH=$1 MPVSS_BENCH_HASH_THREADS=$2 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $1 threads $2 rep $rep', round(d['value']), round(d['compute']['frac'],3))"
done; done
