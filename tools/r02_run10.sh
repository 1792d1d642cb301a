#!/bin/bash
mkdir -p gpurun_out/r02j; O=gpurun_out/r02j
timeout 900 python -m pytest tests/test_gpu_modp.py tests/test_gpu_bench_multirank.py -m gpu -x -q 2>&1 | tail -15
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0"
$B > $O/many.json 2> $O/many.err
MPVSS_DEALER_BUCKETS=0 $B --wb-shares 0 > $O/many_nobuckets.json 2> $O/many_nobuckets.err
MPVSS_BENCH_VERIFY_MANY=0 MPVSS_BENCH_HASH_THREADS=1 $B --wb-shares 0 > $O/chain_d12.json 2> $O/chain_d12.err
MPVSS_BENCH_DEPTH=10 MPVSS_BENCH_VERIFY_MANY=0 MPVSS_BENCH_HASH_THREADS=1 $B --wb-shares 0 > $O/chain_d10.json 2> $O/chain_d10.err
MPVSS_BENCH_DEPTH=8 MPVSS_BENCH_VERIFY_MANY=0 MPVSS_BENCH_HASH_THREADS=1 $B --wb-shares 0 > $O/chain_d8.json 2> $O/chain_d8.err
C2="python bench.py --participants 4096 --threshold 64 --steps 100 --warmup 16 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0"
MPVSS_BENCH_DEPTH=14 $C2 > $O/c2_d14.json 2> $O/c2_d14.err
MPVSS_FD_MIN_SHARES=4096 MPVSS_BENCH_DEPTH=14 $C2 > $O/c2_d14_fd.json 2> $O/c2_d14_fd.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02j/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), {k:round(v,1) for k,v in d['host']['per_box_ms'].items()}, d['compute']['fd_fallbacks'], {k:(round(v) if isinstance(v,float) else v) for k,v in d.get('distribute',{}).items() if k!='note'})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
