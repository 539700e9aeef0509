SPINUP=100 MDQ_CELL_ORDER=conflictfree python3 tools/time_c5.py 128 2>&1 | grep -v "^setup"
SPINUP=100 python3 tools/time_c5.py 128 2>&1 | grep -v "^setup"
SPINUP=100 MDQ_PCG=-2 python3 tools/time_c5.py 128 2>&1 | grep "direct=False"
timeout 900 python -m pytest tests/test_ipcs_gpu.py tests/test_refined_gpu.py -x -q 2>&1 | tail -5
