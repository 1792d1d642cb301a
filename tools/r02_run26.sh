#!/bin/bash
mkdir -p gpurun_out/r02z; O=gpurun_out/r02z
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
MPVSS_BENCH_SMOKE_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 8 --warmup 2 > $O/two_ranks.json 2> $O/two_ranks.err
tail -c 400 $O/two_ranks.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r02z/two_ranks.json') if l.startswith('{"metric"')][-1])
print(round(d['value']), d['n_gpus'], round(d['ms_per_step'],1), d['host'], d['config'])
PY
