#!/bin/bash
mkdir -p gpurun_out/r02l; O=gpurun_out/r02l
timeout 900 python -m pytest tests/test_gpu_modp.py tests/test_gpu_bench_multirank.py tests/test_gpu_robustness.py -m gpu -x -q 2>&1 | tail -8
python tools/bench_dealer.py 65536 16 8 > $O/dealer.txt 2>&1; cat $O/dealer.txt
python tools/bench_dealer.py 65536 16 12 > $O/dealer12.txt 2>&1; cat $O/dealer12.txt
