cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${LOOPS:-4}); do
  MDQ_SHARE_GPU=1 MDQ_DIST_BACKEND=gloo MDQ_BENCH_CPU_LEGS=s2only timeout 500 python bench.py --gpus 2 --share-replay --cpu-budget 1 --steps 6 --warmup 2 --repeats 3 --spinup 60 --s2-steps 20 --envs 16 --s1-steps 4 --s1-warmup 3 --train-steps 3 --s1-solver-steps 200 > gpurun_out/lb_$i.json 2> gpurun_out/lb_$i.err
  python3 - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/lb_$i.json") if l.startswith("{")][0])
    bad={k:v for k,v in d["rates"].items() if isinstance(v,dict) and "error" in v}
    print("run $i", "errors:", json.dumps(bad)[:1500])
except Exception as e:
    print("run $i failed", e); print(open("gpurun_out/lb_$i.err").read()[-1500:])
PY
done
