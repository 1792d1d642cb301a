#!/bin/bash
mkdir -p gpurun_out/r02n; O=gpurun_out/r02n
timeout 1500 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_modp.py tests/test_gpu_ec.py tests/test_gpu_bench_multirank.py -m gpu -x -q 2>&1 | tail -6
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0"
$B > $O/many.json 2> $O/many.err
C2="python bench.py --participants 4096 --threshold 64 --steps 200 --warmup 32 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0 --host-boxes 0"
for d in 14 20 26; do MPVSS_BENCH_DEPTH=$d $C2 > $O/c2_d$d.json 2> $O/c2_d$d.err; done
MPVSS_BENCH_DEPTH=26 MPVSS_BENCH_HASH_THREADS=3 $C2 > $O/c2_d26_h3.json 2> $O/c2_d26_h3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02n/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), round(d['compute']['frac'],3), {k:round(v,1) for k,v in d['host']['per_box_ms'].items()}, d['compute']['fd_fallbacks'], d['host']['slot_init_boxes'], {k:(round(v) if isinstance(v,float) else v) for k,v in d.get('distribute',{}).items() if k!='note'})
        for g,e in d.get('ec',{}).items(): print('   ', g, round(e['value']), round(e['ms_per_box'],2))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
