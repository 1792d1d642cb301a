#!/bin/bash
# repeated full default bench runs: every run is gated on parity (verdicts, dealer digests, CPU sample); prints value / fallbacks
mkdir -p gpurun_out/soak
for i in 1 2 3 4 5 6; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/soak/run$i.json 2> gpurun_out/soak/run$i.err || { echo "run $i FAILED"; tail -5 gpurun_out/soak/run$i.err; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/soak/run$i.json').read().strip().splitlines()[-1])
print($i, round(d['value']), round(d['compute']['frac'],3), 'fd_fallbacks', d['compute']['fd_fallbacks'], 'dealer', round(d['distribute']['value']), 'W_B', round(d['verify_share']['value']), 'ec', [round(e['value']) for e in d['ec'].values()], 'keys', round(d['registered_keys']['value']))
PY
done
