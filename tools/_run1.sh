timeout 900 python -m pytest tests/test_gpu_ec_fd.py tests/test_gpu_host_mirror.py tests/test_gpu_ec.py -x -q -m gpu --durations=5 > gpurun_out/r04_ecq_tests.log 2>&1; tail -12 gpurun_out/r04_ecq_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_quad.json 2> gpurun_out/r04_bench_quad.err; python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_bench_quad.json").read().strip().splitlines()[-1])
print("value",round(d["value"]),"steady",round(d["value_steady_state"]), "verify_share", round(d["verify_share"]["value"]), "extract", round(d["extract_shares"]["value"]), "dist", round(d["distribute"]["value"]), round(d["distribute"]["value_end_to_end"]), "reg", round(d["registered_keys"]["value"]), "host", round(d["host_buffers"]["value"]), {k: round(v["value"]) for k,v in d["configs"].items()})
for g in ("secp256k1","ristretto255"):
    e=d["ec"][g]; print(g, round(e["value"]), round(e["distribute"]["value"]), round(e["distribute"]["value_end_to_end"]), round(e["verify_share"]["value"]), {k: (round(v,2) if isinstance(v,float) else v) for k,v in e["kernel_ms_isolated"].items() if k!="x_path_is"})
print(d.get("secondary_error"))
PY
tail -3 gpurun_out/r04_bench_quad.err
