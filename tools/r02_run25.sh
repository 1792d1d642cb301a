#!/bin/bash
mkdir -p gpurun_out/r02y; O=gpurun_out/r02y
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --ec-boxes 0 --wb-shares 0 --host-boxes 0 --lone-boxes 0 > $O/bench_b.json 2> $O/bench_b.err
python - <<'PY'
import json
for f in ('bench_default','bench_b'):
    d=json.loads(open(f'gpurun_out/r02y/{f}.json').read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), d['roofline'].get('kernel_ms'), (d.get('cpu_baseline') or {}).get('value'))
    for k in ('verify_share','distribute','registered_keys','host_buffers','extract_shares'):
        if k in d: print('  ', k, {a:(round(b) if isinstance(b,float) else b) for a,b in d[k].items() if a!='note'})
    for g,e in d.get('ec',{}).items(): print('   ', g, round(e['value']), round(e['ms_per_box'],2))
PY
