python -m pytest tests/test_ipcs_gpu.py tests/test_golden_gpu.py -m gpu -x -q 2>&1 | tail -2
for W in 768; do MDQ_AT_WG=$W python bench.py --s1-steps 0 --no-cpu-baseline | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($W, r['value'], r['roofline']['kernels_ms_per_step'], r['config']['krylov_iters_per_step'])"; done
cp meshdqn_amd/libtrace.bin meshdqn_amd/libmeshdqn_hip.so; python tools/trace_velocity.py 128 200 | tail -13
