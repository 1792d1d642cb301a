#!/bin/bash
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --ec-boxes 0 --wb-shares 0 --host-boxes 0 --lone-boxes 0"
for rep in a b c d; do for h in 6 8; do
  MPVSS_BENCH_HASH_THREADS=$h $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('threads $h rep $rep', round(d['value']))"
done; done
