exec < /dev/null
timeout 800 python -m pytest tests/test_gpu_fd.py -x -q -m gpu -k "equals_horner" 2>&1 | tail -5 | cut -c1-300
