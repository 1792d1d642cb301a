exec < /dev/null
timeout 800 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "large_threshold" --durations=3 2>&1 | tail -12 | cut -c1-400
