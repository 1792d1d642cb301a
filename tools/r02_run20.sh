#!/bin/bash
mkdir -p gpurun_out/r02t; O=gpurun_out/r02t
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1
tail -16 $O/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02t/bench_default.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), d['roofline'], d['cpu_baseline'])
for k in ('verify_share','distribute','registered_keys','host_buffers'):
    print(k, {a:(round(b) if isinstance(b,float) else b) for a,b in d.get(k,{}).items() if a!='note'})
for g,e in d.get('ec',{}).items(): print('   ', g, round(e['value']), round(e['ms_per_box'],2), e.get('cpu_baseline'))
PY
