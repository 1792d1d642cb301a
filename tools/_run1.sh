timeout 1500 python -m pytest tests/test_gpu_ec_fd.py -x -q -m gpu > gpurun_out/r04_ecq_tests.log 2>&1; tail -15 gpurun_out/r04_ecq_tests.log
python3 tools/ec_x_latency.py > gpurun_out/r04_ec_x_latency.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04_ec_x_latency.txt
