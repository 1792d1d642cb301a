#!/bin/bash
# round-2 measurement batch 2: K7 / two-level-seed tests, A/B of the seeding, chains sweep, CPU-contention experiment
mkdir -p gpurun_out/r02b; O=gpurun_out/r02b
python -m pytest tests/test_gpu_modp.py tests/test_gpu_robustness.py tests/test_gpu_fd.py tests/test_gpu_ec.py tests/test_gpu_extract.py tests/test_gpu_golden.py tests/test_gpu_keyset.py -m gpu -x -q > $O/pytest.log 2>&1
tail -15 $O/pytest.log
B="python bench.py --cpu-sample 0 --wb-shares 0 --registered-keys 0"
for l1 in 1 0; do MPVSS_FD_L1=$l1 $B > $O/l1_$l1.json 2> $O/l1_$l1.err; done
for ch in 4 16 32; do MPVSS_FD_CHAINS=$ch $B > $O/chains_$ch.json 2> $O/chains_$ch.err; done
for d in 6 12; do MPVSS_BENCH_DEPTH=$d $B > $O/depth_$d.json 2> $O/depth_$d.err; done
# CPU contention: every hardware thread of the host busy with somebody else's work
python tools/cpu_hog.py $(nproc) 75 &
sleep 2
$B > $O/hog_many6.json 2> $O/hog_many6.err
MPVSS_BENCH_HASH_THREADS=4 $B > $O/hog_many4.json 2> $O/hog_many4.err
MPVSS_BENCH_VERIFY_MANY=0 MPVSS_BENCH_HASH_THREADS=4 $B > $O/hog_py4.json 2> $O/hog_py4.err
wait
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02b/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), round(d['compute']['modmul_per_share']), {k:round(v,1) for k,v in d['host']['per_box_ms'].items()}, d['compute']['fd_fallbacks'])
    except Exception as e: print(f, 'ERR', e)
PY
