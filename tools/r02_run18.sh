#!/bin/bash
mkdir -p gpurun_out/r02r; O=gpurun_out/r02r
B="python bench.py --gpus 1 --steps 4 --warmup 2 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0"
for one in 3 100; do for l1 in 1 0; do
  MPVSS_EC_ONE_STREAM_FROM=$one MPVSS_EC_FD_L1=$l1 $B > $O/one${one}_l${l1}.json 2> $O/one${one}_l${l1}.err
done; done
MPVSS_EC_ONE_STREAM_FROM=3 MPVSS_BENCH_EC_DEPTH=24 $B > $O/one3_l1_d24.json 2> $O/one3_l1_d24.err
MPVSS_EC_ONE_STREAM_FROM=3 MPVSS_BENCH_EC_DEPTH=10 $B > $O/one3_l1_d10.json 2> $O/one3_l1_d10.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02r/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), [ (g, round(e['value']), round(e['ms_per_box'],2), {k:round(v,1) for k,v in e['host_per_box_ms'].items()}) for g,e in d.get('ec',{}).items()])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
