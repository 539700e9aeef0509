#!/bin/bash
# On the GPU box: regenerate the round-6 artefacts of profiles/ under gpurun_out/<tag>/ (copy them to profiles/ afterwards).
#   tools/refresh_profiles_r06.sh <tag> [quick]
#   quick: kernel trace only (no bench line with the CPU baseline, no PMC passes)
set -u
TAG=${1:-r06}
QUICK=${2:-}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
if [ -z "$QUICK" ]; then
  # (the stdout line is the contract line; the full result - prose, side tables, per-repeat times - is the detail file)
  cd $R && MDQ_BENCH_DETAIL=$OUT/r06_bench_detail.json python3 bench.py > $OUT/r06_bench.json 2> $OUT/bench.err
fi
cd /tmp && export TMPDIR=/tmp
export MDQ_BENCH_DETAIL=$OUT/scratch_detail.json      # (the profiled runs' detail files are not kept)
# kernel trace of the bench command (headline S3 rollouts + S1 + learning loops; C2 / C3 / C5 / deploy side measurements too)
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu-baseline > $OUT/r06_bench_profiled.json 2> $OUT/kt.log
if [ -z "$QUICK" ]; then
  # HBM traffic of the dominant kernels (separate --pmc passes, kernel trace off): the S3 rollout only, then C5 only
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py --no-cpu-baseline --train-steps 0 --no-configs --s1-steps 0 --s2-steps 20 --spinup 20 --repeats 2 > $OUT/$C.log 2>&1
    timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/C5_$C -- python3 $R/tools/time_c5.py pmc > $OUT/C5_$C.log 2>&1
  done
fi
# timelines of one batched env step (S3: both streams; S1), from the rollout timing tool
for F in 1 0; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl$F -- python3 $R/tools/time_rollout.py 128 $F 30 2 > $OUT/tl$F.log 2>&1
  python3 $R/tools/timeline_step.py $OUT/tl$F smooth_linear_kernel -3 > $OUT/r06_timeline_s$((1 + 2 * F))_step.txt 2>&1
done
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, json, shutil, sys
out = sys.argv[1]
for f in glob.glob(f"{out}/kt/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, f"{out}/r06_bench_kernel_stats.csv")
per = {}
for f in glob.glob(f"{out}/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
summ = {k: dict(launches=len(v), mean_us=sum(v) / len(v), min_us=min(v), max_us=max(v)) for k, v in per.items() if "mdq" in k}
json.dump(summ, open(f"{out}/r06_kernel_trace_summary.json", "w"), indent=1)
res = {}
for tag, pre in (("s3", ""), ("c5", "C5_")):
    res[tag] = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = {}
        for f in glob.glob(f"{out}/{pre}{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and "mdq" in r["Kernel_Name"]:
                    vals.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
        res[tag][c] = {k: dict(n=len(v), mean=sum(v) / len(v)) for k, v in vals.items()}
pmc = dict(units="FETCH_SIZE / WRITE_SIZE as reported by rocprofv3 (KiB per launch); hbm bytes = KiB x 1024; the gfx950 x2 correction of the "
                 "guide applies to wide (16 B per lane) coalesced reads: both the raw and the corrected figure are given", per_kernel=res)
for tag in res:
    for k in res[tag].get("FETCH_SIZE", {}):
        f_ = res[tag]["FETCH_SIZE"][k]["mean"]
        w_ = res[tag].get("WRITE_SIZE", {}).get(k, dict(mean=0.0))["mean"]
        pmc.setdefault("hbm_bytes_per_launch", {}).setdefault(tag, {})[k] = dict(raw=(f_ + w_) * 1024, corrected=(2 * f_ + w_) * 1024)
# stamp: the counters describe THESE kernel sources; bench.py refuses the traffic figure when a source has changed since
import hashlib, os
root = os.environ.get("GRAFT_REPO_ROOT", ".")
# (since round 5: EVERY source of the library; bench.py refuses a figure PER KERNEL when one of that kernel's sources has changed)
csrc = os.path.join(root, "meshdqn_amd/csrc")
pmc["kernel_sources_sha256"] = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()
                                for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))}
pmc["passes"] = "ONE collection per committed file: FETCH_SIZE and WRITE_SIZE passes of this script run (s3: the headline rollouts; c5: tools/time_c5.py pmc)"
pmc["git_commit"] = os.environ.get("MDQ_GIT_COMMIT")   # passed in by the caller (the GPU box has no .git)
json.dump(pmc, open(f"{out}/r06_pmc_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summ.items()}, indent=1)[:6000])
print(json.dumps(pmc.get("hbm_bytes_per_launch", {}), indent=1)[:3000])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
ls $OUT
