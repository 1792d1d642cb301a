exec < /dev/null
timeout 1500 bash tools/run_profiles.sh > gpurun_out/run_profiles.log 2>&1
tail -3 gpurun_out/run_profiles.log | cut -c1-300
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof/bench_full_line.json 2> gpurun_out/prof/bench_full_err.txt
tail -2 gpurun_out/prof/bench_full_err.txt | cut -c1-300
python3 -c "
import json
d=json.load(open('gpurun_out/prof/bench_full_line.json'))
print(d['value'], d['ms_per_step'], d.get('secondary_error'))
print({k:(round(v['value']) if isinstance(v,dict) and 'value' in v else None) for k,v in d.items() if isinstance(v,dict)})
print({k:round(v['value']) for k,v in d.get('configs',{}).items()})
"
ls gpurun_out/prof | head -30
