# kernel-level breakdown of the reproducible mode (three launches per step) and of mode 3 at the same tolerance, 45 meshes
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/m2 -- python3 $GRAFT_REPO_ROOT/tools/time_mode2.py > $GRAFT_REPO_ROOT/gpurun_out/m2.log 2>&1
grep "ms per step" $GRAFT_REPO_ROOT/gpurun_out/m2.log
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
for f in glob.glob(f"{R}/gpurun_out/m2/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
find $GRAFT_REPO_ROOT/gpurun_out/m2 -name "*kernel_trace.csv" -delete
