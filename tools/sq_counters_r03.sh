#!/bin/bash
# On the GPU box: SQ / LDS counters of the kernels of the S3 env step, one `--pmc` pass per counter (kernel trace off),
# summarised per kernel into gpurun_out/<tag>/r03_sq_summary.json.     tools/sq_counters_r03.sh <tag>
set -u
TAG=${1:-r03sq}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in LdsBankConflict LdsUtil VALUBusy MemUnitStalled SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py --no-cpu-baseline --train-steps 0 --no-configs --s1-steps 0 --s2-steps 20 --spinup 20 --repeats 2 > $OUT/$C.log 2>&1
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(f"{out}/*/")):
    c = os.path.basename(d.rstrip("/"))
    vals = {}
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "mdq" in r["Kernel_Name"]:
                vals.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        res.setdefault(k, {})[c] = dict(n=len(v), mean=sum(v) / len(v))
json.dump(dict(what="rocprofv3 --pmc <counter> (one pass per counter) of `bench.py --no-cpu-baseline --train-steps 0 --no-configs --s1-steps 0 "
                    "--s2-steps 20 --spinup 20 --repeats 2`: mean per launch and kernel (derived metrics in the tool's units: percent)",
               per_kernel=res), open(f"{out}/r03_sq_summary.json", "w"), indent=1)
for k, v in res.items():
    print(k[:60], {c: round(x["mean"], 2) for c, x in v.items()})
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
