cd /tmp && export TMPDIR=/tmp
export MDQ_TOOL_SOLVER_STEPS=50
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ref_s3 -- python3 $GRAFT_REPO_ROOT/tools/time_rollout.py 128 1 10 2 oracle_stock_ys930_refined > $GRAFT_REPO_ROOT/gpurun_out/ref_s3.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/ref_s3.log
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(f"{R}/gpurun_out/ref_s3/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:18]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
find $GRAFT_REPO_ROOT/gpurun_out/ref_s3 -name "*kernel_trace.csv" -delete
