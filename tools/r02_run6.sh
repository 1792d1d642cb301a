#!/bin/bash
mkdir -p gpurun_out/r02f; O=gpurun_out/r02f
timeout 1200 python -m pytest tests/test_gpu_modp.py tests/test_gpu_bench_multirank.py tests/test_gpu_robustness.py -m gpu -x -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
tail -c 1500 $O/bench_full.err
bash tools/run_profiles.sh > $O/profiles.log 2>&1
tail -5 $O/profiles.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02f/bench_full.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), d['compute']['frac'], d['host'])
for k in ('ec','verify_share','distribute','registered_keys','cpu_baseline'):
    v=d.get(k)
    if k=='ec':
        for g,e in v.items(): print(g, round(e['value']), round(e['ms_per_box'],2), e.get('cpu_baseline',{}).get('value'), e['roofline']['kernel_ms'])
    elif v: print(k, {a:b for a,b in v.items() if a not in ('note','sample')})
PY
