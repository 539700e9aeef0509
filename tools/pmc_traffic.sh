#!/bin/bash
# HBM traffic of evolve_kernel from PMC counters (separate passes, no tracing domains), on the GPU box:
#   tools/pmc_traffic.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/$C.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if ("evolve_" in name or "at_" in name) and r["Counter_Name"] == c:
                key = name.split("(")[0].replace("void ", "")
                per.setdefault(key, []).append(float(r["Counter_Value"]))
    res[c] = {k: dict(n=len(v), mean=sum(v) / len(v), last100_mean=sum(v[-100:]) / len(v[-100:])) for k, v in per.items()}
print(json.dumps(res))
json.dump(res, open(f"{out}/pmc_summary.json", "w"), indent=1)
PY
