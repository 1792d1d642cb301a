# usage: tools/sweep_sched.sh  -- scheduling knobs of the verify-block pipeline on one GPU
for cfg in "32 4 100 0" "16 4 100 0" "8 4 100 0" "16 5 100 0" "8 5 100 0" "32 3 100 0"; do
  set -- $cfg
  echo "chains=$1 depth=$2 a2first=$3 waitseeds=$4"
  MPVSS_FD_CHAINS=$1 MPVSS_BENCH_DEPTH=$2 MPVSS_A2_FIRST_PERCENT=$3 MPVSS_A2_WAIT_SEEDS=$4 python bench.py --steps 12 --warmup 3 --cpu-sample 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['modmul_per_share']), round(d['compute']['frac'],3))"
done
