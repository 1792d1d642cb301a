exec < /dev/null
mkdir -p gpurun_out/r03_fdpair
rm -f gpurun_out/r03_fdpair/t512.txt
for rep in 1 2; do
for v in 512 100000; do
MPVSS_FD_PAIR_MIN_T=$v timeout 400 python3 bench.py --gpus 1 --steps 12 --warmup 3 --threshold 512 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --lone-boxes 0 --config-boxes 0 2>gpurun_out/r03_fdpair/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('t=512 pair_min_t $v', round(d['value']), round(d['ms_per_step'], 2), d.get('secondary_error'))" | tee -a gpurun_out/r03_fdpair/t512.txt
done
done
