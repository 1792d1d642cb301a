#!/bin/bash
mkdir -p gpurun_out/r02x; O=gpurun_out/r02x
timeout 1500 python -m pytest tests/test_gpu_ec.py tests/test_gpu_ec_fd.py tests/test_gpu_golden.py tests/test_gpu_host_mirror.py tests/test_gpu_robustness.py -m gpu -x -q 2>&1 | tail -8
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0"
$B > $O/default.json 2> $O/default.err
MPVSS_EC_X_BATCH=1 MPVSS_BENCH_EC_DEPTH=16 MPVSS_BENCH_EC_HASH_THREADS=3 $B > $O/nobatch.json 2> $O/nobatch.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02x/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['compute']['frac'],3), [ (g, round(e['value']), round(e['ms_per_box'],2), {k:round(v,1) for k,v in e['host_per_box_ms'].items()}) for g,e in d.get('ec',{}).items()])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
