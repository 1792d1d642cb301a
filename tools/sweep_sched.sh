for cfg in "4 3" "8 3" "16 3" "16 4" "24 4"; do
  set -- $cfg
  echo "hwq=$1 depth=$2"
  GPU_MAX_HW_QUEUES=$1 MPVSS_BENCH_DEPTH=$2 python bench.py --steps 8 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
