#!/bin/bash
mkdir -p gpurun_out/r02k; O=gpurun_out/r02k
timeout 1500 python -m pytest tests/test_gpu_robustness.py tests/test_gpu_ec.py tests/test_gpu_ec_fd.py tests/test_gpu_bench_multirank.py tests/test_gpu_extract.py tests/test_gpu_reconstruct.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "c3_c4 or c2" 2>&1 | tail -5
python tools/bench_dealer.py 65536 12 8 > $O/dealer.txt 2>&1; cat $O/dealer.txt
MPVSS_DEALER_BUCKETS=0 python tools/bench_dealer.py 65536 12 8 > $O/dealer_nobuckets.txt 2>&1; cat $O/dealer_nobuckets.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_dealer -o dealer -- python3 $GRAFT_REPO_ROOT/tools/bench_dealer.py 65536 4 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_dealer -name "*kernel_stats*" | head -2 | while read f; do head -12 $f; done
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0"
$B > $O/many.json 2> $O/many.err
MPVSS_BENCH_VERIFY_MANY=0 $B --ec-boxes 0 --host-boxes 0 > $O/pool_h6.json 2> $O/pool_h6.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02k/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), {k:round(v,1) for k,v in d['host']['per_box_ms'].items()}, d['compute']['fd_fallbacks'], {k:(round(v) if isinstance(v,float) else v) for k,v in d.get('distribute',{}).items() if k!='note'}, {k:(round(v) if isinstance(v,float) else v) for k,v in d.get('host_buffers',{}).items() if k!='note'})
        for g,e in d.get('ec',{}).items(): print('   ', g, round(e['value']), round(e['ms_per_box'],2))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
