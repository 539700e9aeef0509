"""GPU: bench.py's contract line, single rank and two torchrun ranks.  The two-rank case shares cuda:0 between the
ranks (MDQ_SHARE_GPU=1) over gloo, because RCCL refuses two ranks on one device: it exercises the multi-rank control
flow (rank-0 build, barriers, max-over-ranks timing, sharded envs, the gradient all-reduce of the learning loop), not
RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "6", "--warmup", "2", "--repeats", "3", "--spinup", "60", "--s2-steps", "20", "--envs", "16",
         "--s1-steps", "4", "--s1-warmup", "3", "--train-steps", "3", "--s1-solver-steps", "200"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline", "roofline_s2_velocity", "rates", "repeats", "value_min",
        "value_max", "s1_env_steps_per_s", "s2_ipcs_env_steps_per_s", "s3_env_steps_per_s", "training_env_steps_per_s"}


def _line(cmd, env):
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_rank_line(lib_built):
    res = _line([sys.executable, "bench.py", "--cpu-budget", "1"] + SMALL, dict(os.environ, MDQ_BENCH_CPU_LEGS="s2only"))
    assert KEYS | {"cpu_baseline"} <= set(res)
    assert res["n_gpus"] == 1 and res["steps"] == 6 and res["warmup"] == 2 and res["scaling"] == "weak"
    # the headline is the S3 env step (median of the repeats); ms_per_step x steps is one timed repeat
    assert res["value"] > 0 and abs(res["value"] - 16 * 6 / (res["ms_per_step"] * 6e-3)) < 1e-6 * res["value"]
    assert res["value"] == res["s3_env_steps_per_s"] == res["rates"]["S3_north_star_step"]["value"]
    assert res["value_min"] <= res["value"] <= res["value_max"]
    assert "S3" in res["config"]["workload"] and res["config"]["krylov_iters_per_ipcs_step"]["velocity_bicgstab"] > 0
    roof = res["roofline"]
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert "smooth_kernel" in roof["kernel"] and roof["launch_ms"] > 0 and roof["step"]["ipcs_leg_bytes_survey_csr_convention"] > 0
    assert res["roofline_s2_velocity"]["bound"] == "lds-atomic/fp64"
    cpu = res["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["s2_ipcs"]["value"] > 0
    for k in ("S1_reference_step", "S2_ipcs_step", "training_loop", "C2_s2_diverged_meshes", "C3_s3_ah93w145", "S3_full_chip",
              "S3_refactorised_pressure", "C5_s2_refined_mesh"):
        assert res["rates"][k]["value"] > 0, (k, res["rates"][k])
    c2 = res["rates"]["C2_s2_diverged_meshes"]     # factors built on the device for every coarsened mesh: no pressure iterations
    assert c2["krylov_iters_per_step"]["pressure"] == 0 and c2["factorisation_status_ok"] and c2["factorisation_ms_per_batch"] > 0
    assert c2["jacobi_cg_variant"]["krylov_iters_per_step"]["pressure"] > 15 and c2["value"] > c2["jacobi_cg_variant"]["value"]
    assert res["rates"]["S3_refactorised_pressure"]["krylov_iters_per_ipcs_step"]["pressure_cg"] == 0
    for k in ("device_loop_s3", "device_loop_s1", "host_loop_s1"):    # learning loop: device-resident (S3, S1) and host-driven
        assert res["rates"]["training_loop"][k]["value"] > 0 and res["rates"]["training_loop"][k]["optimiser_steps"] > 0
    for k in ("s1_env_steps_per_s", "s2_ipcs_env_steps_per_s", "training_env_steps_per_s"):
        assert res[k] > 0


def test_bench_two_ranks_share_one_gpu(lib_built):
    env = dict(os.environ, MDQ_SHARE_GPU="1", MDQ_DIST_BACKEND="gloo")
    res = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                 "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", "--share-replay"] + SMALL, env)
    assert KEYS <= set(res) and "cpu_baseline" not in res          # the CPU leg is rank 0 at N = 1 only
    assert res["n_gpus"] == 2
    assert abs(res["value"] - 2 * 16 * 6 / (res["ms_per_step"] * 6e-3)) < 1e-6 * res["value"]
    assert res["rates"]["training_loop"]["value"] > 0 and res["config"]["collective_backend"]["ranks"] == 2
    for k in ("device_loop_s3", "device_loop_s1", "host_loop_s1"):   # every loop ran (no fallback), with the record all-gather
        r = res["rates"]["training_loop"][k]
        assert "error" not in r and r["optimiser_steps"] > 0 and r["shared_replay"], (k, r)
