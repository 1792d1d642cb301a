#!/bin/bash
# round-2 batch 7: a2 staggering / issue-on-GPU-done A/B at the driver's K=20, W=5; other shapes; fd fault tests
mkdir -p gpurun_out/r02g; O=gpurun_out/r02g
B="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0"
$B > $O/new_a.json 2> $O/new_a.err
MPVSS_A2_CONCURRENCY=0 MPVSS_ISSUE_ON_GPU_DONE=0 $B > $O/old_a.json 2> $O/old_a.err
$B > $O/new_b.json 2> $O/new_b.err
MPVSS_A2_CONCURRENCY=0 MPVSS_ISSUE_ON_GPU_DONE=0 $B > $O/old_b.json 2> $O/old_b.err
MPVSS_A2_CONCURRENCY=1 $B > $O/conc1.json 2> $O/conc1.err
MPVSS_A2_CONCURRENCY=3 $B > $O/conc3.json 2> $O/conc3.err
MPVSS_A2_CONCURRENCY=0 $B > $O/conc0_gpudone.json 2> $O/conc0_gpudone.err
MPVSS_ISSUE_ON_GPU_DONE=0 $B > $O/conc2_nogpudone.json 2> $O/conc2_nogpudone.err
MPVSS_BENCH_DEPTH=6 $B > $O/depth6.json 2> $O/depth6.err
MPVSS_BENCH_DEPTH=12 MPVSS_BENCH_HASH_THREADS=4 $B > $O/depth12.json 2> $O/depth12.err
python bench.py --gpus 1 --steps 40 --warmup 5 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0 > $O/k40.json 2> $O/k40.err
python bench.py --participants 4096 --threshold 64 --steps 60 --warmup 8 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0 > $O/c2.json 2> $O/c2.err
python bench.py --participants 131072 --threshold 1024 --steps 10 --warmup 2 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0 > $O/c5slice.json 2> $O/c5slice.err
timeout 900 python -m pytest tests/test_gpu_fd.py tests/test_gpu_robustness.py tests/test_gpu_keyset.py -m gpu -x -q > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02g/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), round(d['compute']['modmul_per_share']), {k:round(v,1) for k,v in d['host']['per_box_ms'].items()}, d['compute']['fd_fallbacks'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
