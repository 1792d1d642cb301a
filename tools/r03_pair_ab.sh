#!/bin/bash
# A/B of the pair-layout a2 kernel inside the full pipeline: boxes in flight
OUT=gpurun_out/r03_pair; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],1), {k:round(v) for k,v in d['compute']['kernel_ms_sums'].items() if k!='note'})" | tee -a $OUT/ab.txt
}
run pair_d12 MPVSS_A2_PAIR=1
run pair_d3 MPVSS_A2_PAIR=1 MPVSS_BENCH_DEPTH=3
run pair_d5 MPVSS_A2_PAIR=1 MPVSS_BENCH_DEPTH=5
run pair_d8 MPVSS_A2_PAIR=1 MPVSS_BENCH_DEPTH=8
run pair_d16 MPVSS_A2_PAIR=1 MPVSS_BENCH_DEPTH=16
run quad_d12 MPVSS_A2_PAIR=0
