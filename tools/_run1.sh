timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r04_suite_quad.log 2>&1; tail -25 gpurun_out/r04_suite_quad.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_quad.json 2> gpurun_out/r04_bench_quad.err; tail -c 600 gpurun_out/r04_bench_quad.json; tail -3 gpurun_out/r04_bench_quad.err
