timeout 900 python -m pytest tests/test_gpu_scalar_device.py -x -q -m gpu 2>&1 | tail -30
mkdir -p gpurun_out/r03_deal
python3 bench.py --gpus 1 --steps 4 --warmup 2 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --host-boxes 0 --config-boxes 0 --lone-boxes 0 2>gpurun_out/r03_deal/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print(json.dumps({k: v for k, v in d['distribute'].items() if k != 'note'}), d.get('secondary_error'))"
tail -5 gpurun_out/r03_deal/err.txt
