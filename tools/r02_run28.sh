#!/bin/bash
B="python bench.py --gpus 1 --steps 5 --warmup 2 --cpu-sample 0 --registered-keys 0 --ec-boxes 0 --wb-shares 0 --host-boxes 0 --lone-boxes 3"
for lc in 0 16 32 64; do
  MPVSS_BENCH_DEPTH=1 MPVSS_FD_LONE_CHAINS=$lc $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lone chains $lc', 'one box at a time:', round(d['value']), round(d['ms_per_step'],1), 'ms', d['compute']['kernel_ms_isolated'], d['compute']['fd_fallbacks'])"
done
timeout 900 python -m pytest tests/test_gpu_modp.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
