timeout 900 python -m pytest tests/test_gpu_ec_fd.py -x -q -m gpu > gpurun_out/r04_ecq_tests.log 2>&1; tail -3 gpurun_out/r04_ecq_tests.log
bash tools/ec_lone_box_trace.sh 1 2>&1 | grep -v "^W2026" > gpurun_out/r04_ec_lone_quad.txt; grep -v copyBuffer gpurun_out/r04_ec_lone_quad.txt | grep "seeds_win\|table_\|step_\|span"
VARIANTS=2:0 python3 tools/ec_x_latency.py | grep -v amdgpu
VARIANTS=2:0 python3 tools/ec_x_latency.py MPVSS_EC_FD_OCT=1 | grep -v amdgpu
