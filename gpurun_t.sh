exec < /dev/null
mkdir -p gpurun_out/r03_fdpair
timeout 1500 python -m pytest tests/test_gpu_fd.py -x -q -m gpu --durations=5 2>&1 | tail -14 | cut -c1-300
