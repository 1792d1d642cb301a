#!/bin/bash
# round-2 batch 5: EC single-add-site kernel, rocprof of the EC kernels, example binaries, wire format, full GPU suite timing
mkdir -p gpurun_out/r02e; O=gpurun_out/r02e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ec.py tests/test_gpu_golden.py tests/test_gpu_host_mirror.py tests/test_gpu_reconstruct.py -m gpu -x -q > $O/pytest_ec.log 2>&1
tail -12 $O/pytest_ec.log
B="python bench.py --steps 2 --warmup 1 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0"
MPVSS_BENCH_EC_DEPTH=12 $B > $O/ec_d12.json 2> $O/ec_d12.err
MPVSS_BENCH_EC_DEPTH=12 MPVSS_EC_FD_CHAINS=32 $B > $O/ec_d12_ch32.json 2> $O/ec_d12_ch32.err
MPVSS_BENCH_EC_DEPTH=12 MPVSS_EC_FD_CHAINS=8 $B > $O/ec_d12_ch8.json 2> $O/ec_d12_ch8.err
MPVSS_BENCH_EC_DEPTH=16 $B > $O/ec_d16.json 2> $O/ec_d16.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ec_prof -- python3 tools/bench_ec.py --steps 2 > $O/ec_prof.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02e/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']))
        for g,e in d.get('ec',{}).items(): print('   ', g, round(e['value']), round(e['ms_per_box'],2), {k:round(v,2) for k,v in e['kernel_ms_isolated'].items()}, {k:round(v,2) for k,v in e['host_per_box_ms'].items()})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
find $O/ec_prof -name "*kernel_stats.csv" | head -1 | xargs head -24
