exec < /dev/null
mkdir -p gpurun_out/r03_final
SECONDS=0; timeout 900 python3 bench.py > gpurun_out/r03_final/bench_default.json 2> gpurun_out/r03_final/bench_default_err.txt
echo "bench wall seconds: $SECONDS"
python3 -c "
import json
d=json.load(open('gpurun_out/r03_final/bench_default.json'))
print(d['value'], d['steps'], d['warmup'], d['ms_per_step'], d.get('secondary_error'))
print({k:round(v['value']) for k,v in d.get('configs',{}).items()}, round(d['host_buffers']['value']), round(d['distribute']['value']), round(d['distribute']['value_end_to_end']), round(d['extract_shares']['value']), {g:round(d['ec'][g]['value']) for g in d['ec']})
"
