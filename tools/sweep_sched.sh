# usage: tools/sweep_sched.sh  -- short timed regions: box-ordered wide launches (MPVSS_WIDE_FIFO=1) or shared chip
for cfg in "0 5" "1 5" "0 10" "1 10" "0 16" "1 16"; do
  set -- $cfg
  echo "fifo=$1 steps=$2"
  MPVSS_WIDE_FIFO=$1 python bench.py --steps $2 --warmup 2 --cpu-sample 0 --wb-shares 0 --registered-keys 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],1))"
done
