#!/bin/bash
# A/B of the K = 20 tail: hash threads and boxes in flight (headline only, secondary figures off)
OUT=gpurun_out/r03_tail; mkdir -p $OUT
run() { # name, env...
  name=$1; shift
  for rep in 1 2; do
    env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), d['host']['per_box_ms'])" | tee -a $OUT/ab.txt
  done
}
run base
run h8 MPVSS_BENCH_HASH_THREADS=8
run d8h8 MPVSS_BENCH_DEPTH=8 MPVSS_BENCH_HASH_THREADS=8
run d6h8 MPVSS_BENCH_DEPTH=6 MPVSS_BENCH_HASH_THREADS=8
run d16h8 MPVSS_BENCH_DEPTH=16 MPVSS_BENCH_HASH_THREADS=8
run d10h8 MPVSS_BENCH_DEPTH=10 MPVSS_BENCH_HASH_THREADS=8
python3 bench.py --gpus 1 --steps 100 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('k100', round(d['value']), round(d['ms_per_step'],2))" | tee -a $OUT/ab.txt
