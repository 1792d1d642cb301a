#!/bin/bash
mkdir -p gpurun_out/r02p; O=gpurun_out/r02p
timeout 1500 python -m pytest tests/test_gpu_ec.py tests/test_gpu_ec_fd.py tests/test_gpu_golden.py tests/test_gpu_host_mirror.py -m gpu -x -q 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "c3_c4" 2>&1 | tail -4
B="python bench.py --gpus 1 --steps 4 --warmup 2 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0"
$B > $O/l1.json 2> $O/l1.err
MPVSS_EC_FD_L1=0 $B > $O/l0.json 2> $O/l0.err
MPVSS_EC_FD_CHAINS=32 $B > $O/l1_s32.json 2> $O/l1_s32.err
MPVSS_EC_FD_CHAINS=8 $B > $O/l1_s8.json 2> $O/l1_s8.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02p/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']))
        for g,e in d.get('ec',{}).items(): print('   ', g, round(e['value']), round(e['ms_per_box'],2), e['kernel_ms_isolated'], e['host_per_box_ms'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
