mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_ipcs_gpu.py -x -q -k "team or two_workgroups" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -5
