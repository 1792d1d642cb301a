# usage: tools/sweep_sched.sh  -- hardware queue count under the current scheduling
for q in 4 6 8 10 12; do
  echo "hwq=$q"
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 4 --cpu-sample 0 --wb-shares 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), 'keys:', round(d['registered_keys']['value']), round(d['registered_keys']['ms_per_step'],1))"
done
