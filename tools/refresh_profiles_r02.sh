#!/bin/bash
# On the GPU box: regenerate the round-2 artefacts of profiles/ under gpurun_out/<tag>/ (copy them to profiles/ afterwards).
#   tools/refresh_profiles_r02.sh <tag>
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R && python3 bench.py > $OUT/r02_bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --train-steps 0 --no-configs"
# (the kernel trace includes the learning loops: gcn_train_kernel, replay / Adam kernels)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline --no-configs > $OUT/r02_bench_profiled.json 2> $OUT/kt.log
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py $ARGS --s1-steps 0 --s2-steps 20 --spinup 20 --repeats 2 > $OUT/$C.log 2>&1
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, json, shutil, sys
out = sys.argv[1]
# per-kernel stats of the kernel-trace run
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"{out}/r02_bench_kernel_stats.csv")
per = {}
for f in glob.glob(f"{out}/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
summ = {k: dict(launches=len(v), mean_us=sum(v) / len(v), min_us=min(v), max_us=max(v)) for k, v in per.items() if "mdq" in k}
json.dump(summ, open(f"{out}/r02_kernel_trace_summary.json", "w"), indent=1)
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = {}
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "mdq" in r["Kernel_Name"]:
                vals.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
    res[c] = {k: dict(n=len(v), mean=sum(v) / len(v)) for k, v in vals.items()}
sm = [k for k in res["FETCH_SIZE"] if "smooth_kernel" in k]
pmc = dict(units="FETCH_SIZE / WRITE_SIZE as reported by rocprofv3 (KiB); hbm bytes = KiB x 1024; the gfx950 x2 correction of the guide "
                 "applies to wide (16 B per lane) coalesced reads: the smoothing kernel reads coordinates that way and cells as "
                 "4-byte loads, so both the raw and the corrected figure are given", per_kernel=res)
if sm:
    f_, w_ = res["FETCH_SIZE"][sm[0]]["mean"], res["WRITE_SIZE"].get(sm[0], dict(mean=0.0))["mean"]
    pmc.update(kernel=sm[0], fetch_KiB_per_launch=f_, write_KiB_per_launch=w_, hbm_bytes_per_launch_raw=(f_ + w_) * 1024,
               hbm_bytes_per_launch=(2 * f_ + w_) * 1024)
json.dump(pmc, open(f"{out}/r02_smooth_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in pmc.items() if k != "per_kernel"}, indent=1))
print(json.dumps({k: v for k, v in summ.items() if "smooth" in k or "topology" in k or "gcn" in k or "replay" in k or "adam" in k}, indent=1))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls $OUT
