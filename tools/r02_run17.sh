#!/bin/bash
mkdir -p gpurun_out/r02q; O=gpurun_out/r02q
B="python bench.py --gpus 1 --steps 6 --warmup 2 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0"
for q in 8 16 32; do
  GPU_MAX_HW_QUEUES=$q MPVSS_EC_FD_L1=1 $B > $O/q${q}_l1.json 2> $O/q${q}_l1.err
  GPU_MAX_HW_QUEUES=$q MPVSS_EC_FD_L1=0 $B > $O/q${q}_l0.json 2> $O/q${q}_l0.err
done
GPU_MAX_HW_QUEUES=32 MPVSS_BENCH_EC_DEPTH=24 $B > $O/q32_l1_d24.json 2> $O/q32_l1_d24.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02q/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), {k:(round(v) if isinstance(v,float) else v) for k,v in d.get('distribute',{}).items() if k in('value',)}, [ (g, round(e['value']), round(e['ms_per_box'],2)) for g,e in d.get('ec',{}).items()])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
