timeout 900 python -m pytest tests/test_gpu_ec.py tests/test_gpu_ec_deal.py tests/test_gpu_ec_fd.py -x -q -m gpu > gpurun_out/r04_ecq_tests.log 2>&1; tail -3 gpurun_out/r04_ecq_tests.log
bash tools/ec_lone_box_trace.sh 1 2>&1 | grep -v "^W2026" > gpurun_out/r04_ec_lone_quad.txt; grep -v copyBuffer gpurun_out/r04_ec_lone_quad.txt | grep -v "^$"
