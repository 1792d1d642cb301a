exec < /dev/null
mkdir -p gpurun_out/r03_c5
timeout 1200 python -m pytest tests/test_gpu_bench_multirank.py -x -q -m gpu 2>&1 | tail -30 | cut -c1-600 > gpurun_out/r03_c5/mr_final.txt
tail -3 gpurun_out/r03_c5/mr_final.txt
# 4 ranks on the one GPU with the DEFAULT depth logic (no MPVSS_BENCH_DEPTH): small boxes
MPVSS_BENCH_SMOKE_ONE_GPU=1 timeout 600 python3 bench.py --gpus 4 --steps 6 --warmup 2 --participants 4096 --threshold 64 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0 --host-boxes 0 --config-boxes 0 2>gpurun_out/r03_c5/err4.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['n_gpus'], round(d['value']), d['host']['boxes_in_flight'], d['rccl'], d.get('secondary_error'))"
tail -3 gpurun_out/r03_c5/err4.txt | cut -c1-300
