# usage: tools/sweep_sched.sh  -- config[1] (n=4096, t=64): forward differences on/off for small boxes, queues, depth
for cfg in "2048 16 16" "8192 16 16" "2048 8 8" "8192 8 8" "8192 16 12"; do
  set -- $cfg
  echo "fd_min_shares=$1 hwq=$2 depth=$3"
  MPVSS_FD_MIN_SHARES=$1 GPU_MAX_HW_QUEUES=$2 MPVSS_BENCH_DEPTH=$3 python bench.py --participants 4096 --threshold 64 --steps 96 --warmup 16 --cpu-sample 0 --wb-shares 0 --registered-keys 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['compute']['frac'],3), round(d['compute']['modmul_per_share']))"
done
