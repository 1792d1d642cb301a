#!/bin/bash
# round-2 measurement batch 3: full bench line (ec objects, slot init), retire-step A/B, EC kernel profile, full GPU suite
mkdir -p gpurun_out/r02c; O=gpurun_out/r02c
export TMPDIR=/tmp
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
B="python bench.py --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0"
MPVSS_HIP_LIB=$PWD/mpvss_rs_amd/variants/libmpvss_hip_retire0.so $B > $O/retire0.json 2> $O/retire0.err
$B > $O/retire1.json 2> $O/retire1.err
MPVSS_HIP_LIB=$PWD/mpvss_rs_amd/variants/libmpvss_hip_retire0.so $B > $O/retire0_b.json 2> $O/retire0_b.err
$B > $O/retire1_b.json 2> $O/retire1_b.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ec_prof -- python3 tools/bench_ec.py --steps 2 > $O/ec_prof.log 2>&1
python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.log 2>&1
tail -22 $O/pytest_gpu.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02c/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), round(d['compute']['modmul_per_share']), {k:round(v,1) for k,v in d['host']['per_box_ms'].items()})
        if 'ec' in d:
            for g,e in d['ec'].items(): print('   ', g, round(e['value']), round(e['ms_per_box'],1), e['roofline']['kernel_ms'], e['roofline']['x_path_ms'], e.get('cpu_baseline',{}).get('value'))
        for k in ('verify_share','registered_keys','distribute'):
            if k in d: print('   ', k, {a:b for a,b in d[k].items() if a!='note'})
    except Exception as e: print(f, 'ERR', e)
PY
find $O/ec_prof -name "*kernel_stats.csv" | head -1 | xargs head -30
