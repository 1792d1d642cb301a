# usage: tools/sweep_sched.sh  -- boxes in flight, hash threads, chains after the squaring change
for cfg in "8 3 8" "12 4 8" "12 4 16" "16 4 8" "8 3 16" "12 3 12" "8 3 4"; do
  set -- $cfg
  echo "depth=$1 hash_threads=$2 chains=$3"
  MPVSS_BENCH_DEPTH=$1 MPVSS_BENCH_HASH_THREADS=$2 MPVSS_FD_CHAINS=$3 python bench.py --steps 40 --warmup 4 --cpu-sample 0 --wb-shares 0 --registered-keys 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), round(d['compute']['modmul_per_share']), d['host']['absorb_wait_plus_sha256_ms'])"
done
