#!/bin/bash
mkdir -p gpurun_out/r02h; O=gpurun_out/r02h
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0"
$B > $O/d8_h6_a.json 2> $O/d8_h6_a.err
for d in 6 10 12; do MPVSS_BENCH_DEPTH=$d $B > $O/d${d}_h6.json 2> $O/d${d}_h6.err; done
MPVSS_BENCH_HASH_THREADS=8 $B > $O/d8_h8.json 2> $O/d8_h8.err
MPVSS_BENCH_HASH_THREADS=4 $B > $O/d8_h4.json 2> $O/d8_h4.err
MPVSS_BENCH_DEPTH=10 MPVSS_BENCH_HASH_THREADS=8 $B > $O/d10_h8.json 2> $O/d10_h8.err
$B > $O/d8_h6_b.json 2> $O/d8_h6_b.err
MPVSS_FD_CHAINS=4 $B > $O/chains4.json 2> $O/chains4.err
MPVSS_FD_CHAINS=16 $B > $O/chains16.json 2> $O/chains16.err
python bench.py --participants 4096 --threshold 64 --steps 100 --warmup 16 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0 > $O/c2.json 2> $O/c2.err
MPVSS_BENCH_DEPTH=14 python bench.py --participants 4096 --threshold 64 --steps 100 --warmup 16 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0 > $O/c2_d14.json 2> $O/c2_d14.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02h/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), {k:round(v,1) for k,v in d['host']['per_box_ms'].items()}, d['compute']['fd_fallbacks'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
