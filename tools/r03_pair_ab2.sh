#!/bin/bash
# A/B of the pair-layout a2 kernel inside the full pipeline (variants built with PAIR_EXTRA=...)
OUT=gpurun_out/r03_pair; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],1), {k:round(v) for k,v in d['compute']['kernel_ms_sums'].items() if k!='note'})" | tee -a $OUT/ab4.txt
}
run pair_w1_regs MPVSS_A2_PAIR=2 MPVSS_HIP_LIB=$PWD/mpvss_rs_amd/variants/libmpvss_hip_w1.so
run pair_w1_lds MPVSS_A2_PAIR=2 MPVSS_HIP_LIB=$PWD/mpvss_rs_amd/variants/libmpvss_hip_w1lds.so
run pair_w1_regs_d16 MPVSS_A2_PAIR=2 MPVSS_BENCH_DEPTH=16 MPVSS_HIP_LIB=$PWD/mpvss_rs_amd/variants/libmpvss_hip_w1.so
run pair_w1_lds_d16 MPVSS_A2_PAIR=2 MPVSS_BENCH_DEPTH=16 MPVSS_HIP_LIB=$PWD/mpvss_rs_amd/variants/libmpvss_hip_w1lds.so
run quad MPVSS_A2_PAIR=0
