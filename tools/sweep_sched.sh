# usage: tools/sweep_sched.sh  -- number of forward-difference chains under the current scheduling
for c in 16 12 8 6 4; do
  echo "chains=$c"
  MPVSS_FD_CHAINS=$c python bench.py --steps 32 --warmup 4 --cpu-sample 0 --wb-shares 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), round(d['compute']['modmul_per_share']), 'keys:', round(d['registered_keys']['value']), round(d['registered_keys']['ms_per_step'],1))"
done
