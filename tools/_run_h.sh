timeout 1700 python -m pytest tests/test_refined_gpu.py -x -q -k "twice" 2>&1 | tail -25
timeout 900 python -m pytest tests/test_env_gpu.py tests/test_refined_gpu.py -x -q -k "not twice" 2>&1 | tail -4
