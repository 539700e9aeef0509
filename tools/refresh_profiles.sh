#!/bin/bash
# On the GPU box: regenerate every artefact profiles/README.md lists, under gpurun_out/<tag>/.
#   tools/refresh_profiles.sh <tag>
set -u
TAG=${1:-fin}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s2 -- python3 $R/bench.py --no-cpu-baseline --s1-steps 0 > $OUT/s2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -- python3 $R/bench.py --no-cpu-baseline > $OUT/full.log 2>&1
cd $R
bash tools/pmc_traffic.sh $TAG/pmc --s1-steps 0 > $OUT/pmc.log 2>&1
bash tools/pmc_sq.sh $TAG/sq > $OUT/sq.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
per = {}
for f in glob.glob(f"{out}/s2/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at_" in n:
            per.setdefault(n.split("(")[0].replace("void ", ""), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
res = {}
for k, v in per.items():
    v.sort()
    d = [x[1] / 1e3 for x in v]
    res[k] = dict(launches=len(d), mean_all_us=sum(d) / len(d), mean_last_400_us=sum(d[-400:]) / len(d[-400:]), mean_first_300_us=sum(d[:300]) / len(d[:300]))
json.dump(dict(per_kernel=res), open(f"{out}/s2_kernel_trace_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
# keep the merged output small: drop the raw traces, keep stats + summaries
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
ls -R $OUT | head -50
