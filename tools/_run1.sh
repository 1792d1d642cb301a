run() { # name shape-args env...
  name=$1; shape=$2; shift 2
  env "$@" python3 bench.py --gpus 1 $shape --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 \
      --host-boxes 0 --config-boxes 0 --lone-boxes 1 --steady-steps 0 2>gpurun_out/_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', round(d['value']), round(d['ms_per_step'], 2), {k: round(v) for k, v in d['compute']['kernel_ms_sums'].items() if k != 'note'})" || tail -3 gpurun_out/_err.txt
}
C5="--participants 131072 --threshold 1024 --steps 10 --warmup 3"
HL="--steps 20 --warmup 5"
for rep in 1 2; do
run c5_default "$C5" MPVSS_BENCH_DEPTH=10
run c5_stepprio "$C5" MPVSS_BENCH_DEPTH=10 MPVSS_HIP_LIB=ab_libs/libmpvss_hip_stepprio.so
run hl_default "$HL" X=1
run hl_stepprio "$HL" MPVSS_HIP_LIB=ab_libs/libmpvss_hip_stepprio.so
done
