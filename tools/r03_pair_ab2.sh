#!/bin/bash
OUT=gpurun_out/r03_pair; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],1), {k:round(v) for k,v in d['compute']['kernel_ms_sums'].items() if k!='note'})" | tee -a $OUT/ab2.txt
}
for v in w4pad w2 w8; do run pair_$v MPVSS_A2_PAIR=1 MPVSS_HIP_LIB=$PWD/mpvss_rs_amd/variants/libmpvss_hip_$v.so; done
run pair_w4 MPVSS_A2_PAIR=1
run quad MPVSS_A2_PAIR=0
