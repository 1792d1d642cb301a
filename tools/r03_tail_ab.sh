#!/bin/bash
# A/B of the K = 20 tail: the last boxes of a run enqueued in parts (MPVSS_TAIL_BOXES x MPVSS_TAIL_PARTS), hash threads
OUT=gpurun_out/r03_tail; mkdir -p $OUT
run() { name=$1; shift
  for rep in 1 2 3; do
    env "$@" python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value']), round(d['ms_per_step'],2))" | tee -a $OUT/ab2.txt
  done
}
run parts1 MPVSS_TAIL_PARTS=1
run last1x4 MPVSS_TAIL_BOXES=1

run last12x4 MPVSS_TAIL_BOXES=12
run last12x2 MPVSS_TAIL_BOXES=12 MPVSS_TAIL_PARTS=2

