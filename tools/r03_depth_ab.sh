#!/bin/bash
# boxes in flight / hash threads with the pair a2 kernel (headline only)
OUT=gpurun_out/r03_pair; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), {k:round(v) for k,v in d['compute']['kernel_ms_sums'].items() if k!='note'}, {k:round(v,1) for k,v in d['host']['per_box_ms'].items()})" | tee -a $OUT/depth.txt
}
run d12h6
run d8h6 MPVSS_BENCH_DEPTH=8
run d16h8 MPVSS_BENCH_DEPTH=16 MPVSS_BENCH_HASH_THREADS=8
run d20h8 MPVSS_BENCH_DEPTH=20 MPVSS_BENCH_HASH_THREADS=8
run d12h8 MPVSS_BENCH_HASH_THREADS=8
run d12h6_chains16 MPVSS_FD_CHAINS=16
run d12h6_q16 GPU_MAX_HW_QUEUES=16
python3 bench.py --gpus 1 --steps 100 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('k100', round(d['value']), round(d['ms_per_step'],2))" | tee -a $OUT/depth.txt
