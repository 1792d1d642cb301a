#!/bin/bash
# round-2 batch 4: new EC path tests, EC A/B (windows on/off), depth sweep, bench line
mkdir -p gpurun_out/r02d; O=gpurun_out/r02d
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ec.py tests/test_gpu_ec_fd.py tests/test_gpu_extract.py tests/test_gpu_golden.py tests/test_gpu_host_mirror.py -m gpu -x -q > $O/pytest_ec.log 2>&1
tail -25 $O/pytest_ec.log
timeout 600 python -m pytest "tests/test_gpu_configs.py::test_c3_c4_curve_groups_full_size" tests/test_gpu_modp.py -m gpu -x -q > $O/pytest_cfg.log 2>&1
tail -8 $O/pytest_cfg.log
B="python bench.py --steps 6 --warmup 1 --cpu-sample 0 --registered-keys 0 --lone-boxes 0"
$B > $O/ec_default.json 2> $O/ec_default.err
MPVSS_EC_WINDOWS=0 $B --wb-shares 0 > $O/ec_nowin.json 2> $O/ec_nowin.err
for d in 2 4 12; do MPVSS_BENCH_EC_DEPTH=$d $B --wb-shares 0 > $O/ec_depth$d.json 2> $O/ec_depth$d.err; done
MPVSS_BENCH_EC_HASH_THREADS=6 $B --wb-shares 0 > $O/ec_hash6.json 2> $O/ec_hash6.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02d/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']))
        for g,e in d.get('ec',{}).items(): print('   ', g, round(e['value']), round(e['ms_per_box'],2), {k:round(v,2) for k,v in e['kernel_ms_isolated'].items()}, {k:round(v,2) for k,v in e['host_per_box_ms'].items()})
        if 'verify_share' in d: print('   ', {a:b for a,b in d['verify_share'].items() if a!='note'})
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
