#!/bin/bash
mkdir -p gpurun_out/r02ab; O=gpurun_out/r02ab
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --ec-boxes 0 --wb-shares 0 --host-boxes 0 --lone-boxes 0"
for i in 1 2 3 4 5; do
  MPVSS_TRACE_ENQUEUE=1 $B > $O/t$i.json 2> $O/t$i.err
  python -c "import json; d=json.loads(open('$O/t$i.json').read().strip().splitlines()[-1]); print($i, round(d['value']), d['host']['per_box_ms'])"
  grep -c "slow enqueue" $O/t$i.err; grep "slow enqueue" $O/t$i.err | tail -4
done
