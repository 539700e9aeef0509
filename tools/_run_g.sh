export MDQ_TOOL_SOLVER_STEPS=50
python3 tools/time_rollout.py 128 1 10 3 oracle_stock_ys930_refined 2>&1 | tail -1
MDQ_SETUP_UNSTAGED=1 python3 tools/time_rollout.py 128 1 10 3 oracle_stock_ys930_refined 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_refined_gpu.py tests/test_env_gpu.py -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06p_kt -- python3 $GRAFT_REPO_ROOT/tools/time_rollout.py 128 1 10 2 oracle_stock_ys930_refined > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(f"{R}/gpurun_out/r06p_kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
find $GRAFT_REPO_ROOT/gpurun_out/r06p_kt -name "*kernel_trace.csv" -delete
