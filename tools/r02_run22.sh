#!/bin/bash
mkdir -p gpurun_out/r02v; O=gpurun_out/r02v
B="python bench.py --gpus 1 --steps 3 --warmup 1 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0 --ec-boxes 48"
for cfg in "8 30 6" "16 30 6" "12 30 6" "8 24 6" "16 24 4" "10 30 8"; do
  set -- $cfg
  MPVSS_EC_X_BATCH=$1 MPVSS_BENCH_EC_DEPTH=$2 MPVSS_BENCH_EC_HASH_THREADS=$3 $B > $O/xb$1_d$2_h$3.json 2> $O/xb$1_d$2_h$3.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02v/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), [ (g, round(e['value']), round(e['ms_per_box'],2), {k:round(v,1) for k,v in e['host_per_box_ms'].items()}) for g,e in d.get('ec',{}).items()])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
