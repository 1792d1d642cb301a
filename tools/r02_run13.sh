#!/bin/bash
# C5 slice: where does the enqueue time go?  (HIP API trace, no counters)
mkdir -p gpurun_out/r02m; O=gpurun_out/r02m
export TMPDIR=/tmp
A="bench.py --participants 131072 --threshold 1024 --steps 6 --warmup 2 --cpu-sample 0 --wb-shares 0 --registered-keys 0 --ec-boxes 0 --lone-boxes 0 --host-boxes 0"
python $A > $O/c5.json 2> $O/c5.err
rocprofv3 --hip-trace --stats --output-format csv -d $O/hip -- python3 $A > $O/hip.log 2>&1
python - <<'PY'
import json,glob,csv
d=json.loads(open('gpurun_out/r02m/c5.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],1), round(d['compute']['frac'],3), d['host']['per_box_ms'], d['host']['boxes_in_flight'])
for f in glob.glob('gpurun_out/r02m/hip/**/*hip_api_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    for r in rows[:14]: print(r['Name'], r['Calls'], round(float(r['TotalDurationNs'])/1e6,1),'ms total', round(float(r['AverageNs'])/1e3,1),'us avg', round(float(r['MaxNs'])/1e6,2),'ms max')
PY
