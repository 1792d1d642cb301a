run() { # name env...
  name=$1; shift
  env "$@" MPVSS_BENCH_DEPTH=10 python3 bench.py --gpus 1 --participants 131072 --threshold 1024 --steps 10 --warmup 3 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 \
      --host-boxes 0 --config-boxes 0 --lone-boxes 1 --steady-steps 0 2>gpurun_out/_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['value']), round(d['ms_per_step'], 2), 'fallbacks', d.get('host', {}).get('fd_fallbacks'), {k: round(v) for k, v in d['compute']['kernel_ms_sums'].items() if k != 'note'})" || tail -3 gpurun_out/_err.txt
}
for rep in 1 2; do
run c5_default MPVSS_X_CUS=0
run c5_x64 MPVSS_X_CUS=64
run c5_x96 MPVSS_X_CUS=96
run c5_x32 MPVSS_X_CUS=32
done
