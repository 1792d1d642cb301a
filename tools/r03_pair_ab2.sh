#!/bin/bash
# A/B of the pair-layout kernels inside the full pipeline: MPVSS_PAIR bit mask (1 a2, 2 tables, 4 g^r, 8 a1)
OUT=gpurun_out/r03_pair; mkdir -p $OUT
run() { name=$1; shift
  env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],1), {k:round(v) for k,v in d['compute']['kernel_ms_sums'].items() if k!='note'}, d['compute']['kernel_ms_isolated'])" | tee -a $OUT/ab5.txt
}
python -m pytest tests/test_gpu_modp.py tests/test_gpu_fd.py -m gpu -x -q 2>&1 | tail -2
run pair15 MPVSS_PAIR=15
run pair1 MPVSS_PAIR=1
run pair3 MPVSS_PAIR=3
run pair7 MPVSS_PAIR=7
run pair9 MPVSS_PAIR=9
run pair0 MPVSS_PAIR=0
