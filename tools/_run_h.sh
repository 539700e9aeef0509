timeout 1700 python -m pytest tests/test_refined_gpu.py -x -q -k "twice_refined_mesh_episode" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q -k "single_rank" 2>&1 | tail -5
