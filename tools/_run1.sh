run() { name=$1; shift
  python3 bench.py --gpus 1 --steps 5 --warmup 2 --cpu-sample 0 --host-boxes 0 --config-boxes 0 --registered-keys 0 --steady-steps 0 "$@" 2>gpurun_out/_err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$name', 'verify_share', round(d['verify_share']['value']), 'extract', round(d['extract_shares']['value']), 'distribute', round(d['distribute']['value']), 'e2e', round(d['distribute']['value_end_to_end']))" || tail -3 gpurun_out/_err.txt
}
run no_ec --ec-boxes 0
run with_ec --ec-boxes 64
run no_ec --ec-boxes 0
