#!/bin/bash
# On the GPU box: timeline of one S3 env step on the red-refined ys930 (128 environments): kernel trace of tools/time_rollout.py,
# the per-kernel statistics and the timeline of one step between two smoothing launches.
#   tools/prof_refined_s3_r06.sh <tag>
TAG=${1:-ref_s3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MDQ_TOOL_SOLVER_STEPS=50
python3 $R/tools/time_rollout.py 128 1 10 3 oracle_stock_ys930_refined > $OUT/plain.log 2>&1
tail -1 $OUT/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/time_rollout.py 128 1 10 2 oracle_stock_ys930_refined > $OUT/prof.log 2>&1
tail -1 $OUT/prof.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(f"{out}/kernel_stats.txt", "w") as g:
        for r in rows[:20]:
            line = f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}"
            print(line); g.write(line + "\n")
PY
python3 $R/tools/timeline_step.py $OUT/kt smooth_flow_kernel -3 > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
