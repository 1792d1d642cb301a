exec < /dev/null
mkdir -p gpurun_out/r03_final
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_final/bench_full_line.json 2> gpurun_out/r03_final/bench_full_err.txt
python3 -c "
import json
d=json.load(open('gpurun_out/r03_final/bench_full_line.json'))
print(d['value'], d['ms_per_step'], d.get('secondary_error'), d['roofline']['kernel_ms'], d['compute']['frac'])
print({k:(round(v['value']), round(v['compute']['frac'],3), v['boxes']) for k,v in d.get('configs',{}).items()}, round(d['host_buffers']['value']), round(d['distribute']['value']), round(d['distribute']['value_end_to_end']), round(d['distribute']['value_one_call_host_buffers_end_to_end']), round(d['extract_shares']['value']), {g:round(d['ec'][g]['value']) for g in d['ec']}, round(d['verify_share']['value']), round(d['registered_keys']['value']))
"
