#!/bin/bash
# On the GPU box: per-kernel means of the S1 / S3 env step on ys930 red-refined TWICE (12 924 vertices; the 16 384-vertex instances).
#   tools/prof_refined2_r06.sh <tag> [B]
TAG=${1:-ref2}
B=${2:-32}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MDQ_TOOL_SOLVER_STEPS=10
for F in 0 1; do
  python3 $R/tools/time_rollout.py $B $F 3 2 oracle_stock_ys930_refined2 > $OUT/plain$F.log 2>&1
  tail -1 $OUT/plain$F.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/tools/time_rollout.py $B 1 3 2 oracle_stock_ys930_refined2 > $OUT/prof.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(f"{out}/kernel_stats.txt", "w") as g:
        for r in rows[:14]:
            line = f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}"
            print(line); g.write(line + "\n")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
