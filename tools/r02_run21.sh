#!/bin/bash
mkdir -p gpurun_out/r02u; O=gpurun_out/r02u
timeout 1500 python -m pytest tests/test_gpu_ec.py tests/test_gpu_ec_fd.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -12
B="python bench.py --gpus 1 --steps 4 --warmup 2 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0"
for xb in 4 1 2 8; do MPVSS_EC_X_BATCH=$xb $B > $O/xb$xb.json 2> $O/xb$xb.err; done
MPVSS_EC_X_BATCH=4 MPVSS_BENCH_EC_DEPTH=24 $B > $O/xb4_d24.json 2> $O/xb4_d24.err
MPVSS_EC_X_BATCH=8 MPVSS_BENCH_EC_DEPTH=24 MPVSS_BENCH_EC_HASH_THREADS=5 $B > $O/xb8_d24_h5.json 2> $O/xb8_d24_h5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02u/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), [ (g, round(e['value']), round(e['ms_per_box'],2), {k:round(v,1) for k,v in e['host_per_box_ms'].items()}) for g,e in d.get('ec',{}).items()])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
