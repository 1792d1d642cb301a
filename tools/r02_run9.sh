#!/bin/bash
# full GPU suite + the driver's bench call + the Python-driven (multi-rank style) pipeline on one GPU
mkdir -p gpurun_out/r02i; O=gpurun_out/r02i
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 ) > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0"
MPVSS_BENCH_VERIFY_MANY=0 MPVSS_BENCH_HASH_THREADS=1 $B > $O/py_h1.json 2> $O/py_h1.err
MPVSS_BENCH_VERIFY_MANY=0 $B > $O/py_h6.json 2> $O/py_h6.err
$B > $O/many_b.json 2> $O/many_b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02i/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), {k:round(v,1) for k,v in d['host']['per_box_ms'].items()}, d['compute']['fd_fallbacks'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
