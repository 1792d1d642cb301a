# usage: tools/sweep_sched.sh  -- chains x boxes in flight x hash threads with single-wave workgroups
for cfg in "8 8 3" "16 8 3" "6 8 3" "8 12 4" "8 6 3" "8 8 5"; do
  set -- $cfg
  echo "chains=$1 depth=$2 hash_threads=$3"
  MPVSS_FD_CHAINS=$1 MPVSS_BENCH_DEPTH=$2 MPVSS_BENCH_HASH_THREADS=$3 python bench.py --steps 40 --warmup 4 --cpu-sample 0 --wb-shares 0 --registered-keys 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3))"
done
