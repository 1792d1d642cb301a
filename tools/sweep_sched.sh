# usage: tools/sweep_sched.sh  -- host hash threads x boxes in flight x length of the timed region
for cfg in "3 8 5 2" "4 12 5 2" "4 8 24 4" "4 12 24 4" "2 8 24 4" "3 6 24 4" "3 8 50 4"; do
  set -- $cfg
  echo "hash_threads=$1 depth=$2 steps=$3 warmup=$4"
  MPVSS_BENCH_HASH_THREADS=$1 MPVSS_BENCH_DEPTH=$2 python bench.py --steps $3 --warmup $4 --cpu-sample 0 --wb-shares 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), 'keys:', round(d['registered_keys']['value']), round(d['registered_keys']['ms_per_step'],1))"
done
