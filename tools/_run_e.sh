timeout 1500 python -m pytest tests/test_env_gpu.py tests/test_refined_gpu.py tests/test_stock_gpu.py -x -q 2>&1 | tail -5
export MDQ_TOOL_SOLVER_STEPS=50
python3 tools/time_rollout.py 128 0 10 3 oracle_stock_ys930_refined 2>&1 | tail -1
python3 tools/time_rollout.py 128 1 10 3 oracle_stock_ys930_refined 2>&1 | tail -1
python3 tools/time_rollout.py 128 1 50 3 2>&1 | tail -1
