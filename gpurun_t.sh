exec < /dev/null
mkdir -p gpurun_out/r03_c5
rm -f gpurun_out/r03_c5/depth.txt
export MPVSS_BENCH_CONFIGS=c5_slice
for d in 6 10 14; do
MPVSS_BENCH_C5_DEPTH=$d timeout 400 python3 bench.py --gpus 1 --steps 2 --warmup 1 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --lone-boxes 0 --config-boxes 128 2>gpurun_out/r03_c5/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
c = d['configs']['c5_slice']
print('depth $d', round(c['value']), round(c['ms_per_box'], 1), c['boxes'], round(c['compute']['frac'], 3), c['host_ms_per_box'], d.get('secondary_error'))" | tee -a gpurun_out/r03_c5/depth.txt
done
