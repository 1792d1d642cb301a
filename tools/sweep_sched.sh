# usage: tools/sweep_sched.sh  -- hardware queue count (GPU_MAX_HW_QUEUES) x pipeline depth
for cfg in "8 8" "16 8" "16 16" "32 16"; do
  set -- $cfg
  echo "config1 n=4096 t=64 hwq=$1 depth=$2"
  GPU_MAX_HW_QUEUES=$1 MPVSS_BENCH_DEPTH=$2 python bench.py --participants 4096 --threshold 64 --steps 64 --warmup 16 --cpu-sample 0 --wb-shares 0 --registered-keys 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), round(d['compute']['frac'],3))"
done
for cfg in "8 4" "16 8" "24 12"; do
  set -- $cfg
  echo "headline hwq=$1 depth=$2"
  GPU_MAX_HW_QUEUES=$1 MPVSS_BENCH_DEPTH=$2 python bench.py --steps 16 --warmup 4 --cpu-sample 0 --wb-shares 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), 'keys:', round(d['registered_keys']['value']), round(d['registered_keys']['ms_per_step'],1), round(d['registered_keys']['compute_frac'],3))"
done
