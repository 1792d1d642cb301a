#!/bin/bash
mkdir -p gpurun_out/r02aa; O=gpurun_out/r02aa
timeout 1500 python -m pytest tests/test_gpu_modp.py tests/test_gpu_golden.py tests/test_gpu_robustness.py tests/test_gpu_keyset.py tests/test_gpu_fd.py -m gpu -x -q 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "c2 or headline" 2>&1 | tail -4
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --registered-keys 0 --lone-boxes 0 --wb-shares 0 --host-boxes 0 --ec-boxes 0"
$B > $O/slide_a.json 2> $O/slide_a.err
MPVSS_C_SLIDING=0 $B > $O/fixed_a.json 2> $O/fixed_a.err
$B > $O/slide_b.json 2> $O/slide_b.err
MPVSS_C_SLIDING=0 $B > $O/fixed_b.json 2> $O/fixed_b.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02aa/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],2), round(d['compute']['frac'],3), round(d['compute']['modmul_per_share'],1))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
