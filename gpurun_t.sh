exec < /dev/null
timeout 500 python -m pytest tests/test_gpu_scalar_device.py -x -q -m gpu 2>&1 | tail -12 | cut -c1-400
