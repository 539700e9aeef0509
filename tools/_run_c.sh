python3 tools/check_tilemaps.py 128 2>&1 | grep "maps="
timeout 900 python -m pytest tests/test_ipcs_gpu.py -x -q -k "two_level or polynomial or two_workgroups or team" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_refined_gpu.py -x -q 2>&1 | tail -8
