#!/bin/bash
# SQ counters of the IPCS kernels (own pass, no tracing domains), on the GPU box:  tools/pmc_sq.sh <outdir> [bench args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --s1-steps 0 "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $OUT/p2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --s1-steps 0 "$@" > $OUT/p2.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
res = {}
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "at_" in name:
            key = name.split("(")[0].replace("void ", "")
            res.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
summ = {k: {c: sum(v[-100:]) / len(v[-100:]) for c, v in d.items()} for k, d in res.items()}
print(json.dumps(summ, indent=1))
json.dump(summ, open(f"{out}/sq_summary.json", "w"), indent=1)
PY
