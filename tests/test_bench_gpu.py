"""GPU: bench.py's contract line, single rank and two ranks - started by bench.py itself (`--gpus 2`: the parent starts
the rank processes) and by torchrun (the driver's form).  The two-rank cases share cuda:0 between the ranks
(MDQ_SHARE_GPU=1) over gloo, because RCCL refuses two ranks on one device: they exercise the multi-rank control flow
(rank-0 build, barriers, max-over-ranks timing, sharded envs, the gradient all-reduce and the record all-gather of the
learning loop), not RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "6", "--warmup", "2", "--repeats", "3", "--spinup", "60", "--s2-steps", "20", "--envs", "16",
         "--s1-steps", "4", "--s1-warmup", "3", "--train-steps", "3", "--s1-solver-steps", "200"]
LINE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "rates", "repeats", "value_min", "value_max",
             "steady_state_ms_per_step", "detail"}
KEYS = LINE_KEYS - {"detail"} | {"roofline_s2_velocity", "s1_env_steps_per_s", "s2_ipcs_env_steps_per_s", "s3_env_steps_per_s",
                                 "training_env_steps_per_s", "kernel_scratch"}       # of the detail file


def _line(cmd, env, tmp_path=None):
    """Runs bench.py; returns the DETAIL (the full result, written to a file) after checking the stdout line against it:
    the line is the last thing on stdout, <= 8 KB (the driver reads the tail of stdout), parses on its own, and carries
    numbers only for the side rates."""
    import tempfile
    fd, detail_path = tempfile.mkstemp(prefix="mdq_bench_detail_test_", suffix=".json")
    os.close(fd)
    try:
        out = subprocess.run(cmd, cwd=ROOT, env=dict(env, MDQ_BENCH_DETAIL=detail_path), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1 and out.stdout.rstrip("\n").endswith(lines[0]), out.stdout[-2000:]
        assert len(lines[0]) <= 8192, len(lines[0])
        line = json.loads(out.stdout[-8192:][out.stdout[-8192:].index("{"):])     # what a reader of the last 8 KB of stdout sees
        detail = json.load(open(detail_path))
    finally:
        os.unlink(detail_path)
    assert LINE_KEYS <= set(line) and line["detail"] == os.path.basename(detail_path)
    for k in ("value", "ms_per_step", "value_min", "value_max"):
        assert abs(line[k] - detail[k]) <= 1e-5 * abs(detail[k])
    assert set(line["rates"]) == set(detail["rates"])
    assert all(v is None or v == "error" or isinstance(v, (int, float)) for v in line["rates"].values())
    for k in ("bound", "achieved", "peak", "frac", "traffic", "unit", "kernel", "launch_ms", "floor_ms", "launch_over_floor",
              "algorithmic_bytes_per_launch", "traffic_source"):
        assert k in line["roofline"], k
    assert "cpu_baseline" not in detail or {"value", "cores", "kind", "sample", "s2_ipcs"} <= set(line["cpu_baseline"])
    return detail


def test_bench_single_rank_line(lib_built):
    res = _line([sys.executable, "bench.py", "--cpu-budget", "1"] + SMALL, dict(os.environ, MDQ_BENCH_CPU_LEGS="s2only"))
    assert KEYS | {"cpu_baseline"} <= set(res)
    assert res["n_gpus"] == 1 and res["steps"] == 6 and res["warmup"] == 2 and res["scaling"] == "weak"
    # the headline is the S3 env step (median of the repeats); ms_per_step x steps is one timed repeat
    assert res["value"] > 0 and abs(res["value"] - 16 * 6 / (res["ms_per_step"] * 6e-3)) < 1e-6 * res["value"]
    assert res["value"] == res["s3_env_steps_per_s"] == res["rates"]["S3_north_star_step"]["value"]
    assert res["value_min"] <= res["value"] <= res["value_max"]
    assert 0 < res["steady_state_ms_per_step"] <= res["ms_per_step"] * 1.5          # the HIP-event period of a step inside the rollouts
    ks = res["kernel_scratch"]
    # (with a private segment: the tile kernels that call the matrix-on-chip pressure solver, round 6: a `noinline` function of 256
    #  VGPRs - its callee saves, the caller's saves around the call, and one 7-row vector it keeps there; nothing inside the velocity
    #  loops: checked by source line in the ISA.  `topology_kernel<4>`: 0 B since round 6 - 344 B in round 4, 124 B in round 5;
    #  `topology_kernel<16>`: the 16 384-vertex coverage instance, 1 024 threads at 128 VGPRs - 1 256 B, not on any timed path)
    big = ("evolve_kernel<5", "evolve_team_tiles_kernel<false>", "topology_kernel<16>")
    assert ks["kernels"] >= 55 and all(v["scratch"] == 0 for k, v in ks["hot_kernels"].items() if not any(b in k for b in big)), ks
    assert "S3" in res["config"]["workload"] and res["config"]["krylov_iters_per_ipcs_step"]["velocity_bicgstab"] > 0
    roof = res["roofline"]
    assert roof["bound"] == "hbm" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert 0.05 < roof["floor_ms"] < roof["launch_ms"] and abs(roof["launch_over_floor"] - roof["launch_ms"] / roof["floor_ms"]) < 1e-9
    assert "smooth_linear_kernel" in roof["kernel"] and roof["launch_ms"] > 0 and roof["step_survey_convention_equivalent"]["ipcs_leg_bytes_survey_csr_convention"] > 0
    assert roof["traffic_measured_in_run"] is False and "step" not in roof
    assert res["roofline_s2_velocity"]["bound"] == "lds-atomic/fp64"
    cpu = res["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["s2_ipcs"]["value"] > 0
    dep = res["rates"]["deploy_episode_s"]
    assert dep["value"] > 0 and dep["resimulated_meshes"] == 45 and dep["same_first_rows"] and dep["sequential_3_removals_s"] > 0
    assert isinstance(res["stream_setup"], list) and res["stream_setup"]
    for k in ("S1_reference_step", "S2_ipcs_step", "training_loop", "C2_s2_diverged_meshes", "C3_s3_ah93w145", "S3_full_chip",
              "S3_refactorised_pressure", "C5_s2_refined_mesh", "C5_s1_refined_mesh", "C5_s3_refined_mesh", "C5b_s1_twice_refined"):
        assert res["rates"][k].get("value", 0) > 0, (k, res["rates"][k])
    r5 = res["rates"]["C5_s2_refined_mesh"]["roofline"]        # the HBM-bound kernel of the path: its own roofline numbers
    assert r5["bound"] == "hbm" and "evolve_team_tiles" in r5["kernel"] and 0.05 < r5["frac"] < 1.0
    assert abs(r5["frac"] - r5["achieved"] / r5["peak"]) < 1e-12 and r5["algorithmic_bytes_per_launch"] > 1e8
    c2 = res["rates"]["C2_s2_diverged_meshes"]     # factors built on the device for every coarsened mesh: no pressure iterations
    assert c2["krylov_iters_per_step"]["pressure"] == 0 and c2["factorisation_status_ok"] and c2["factorisation_ms_per_batch"] > 0
    assert c2["jacobi_cg_variant"]["krylov_iters_per_step"]["pressure"] > 15 and c2["value"] > c2["jacobi_cg_variant"]["value"]
    assert res["rates"]["S3_refactorised_pressure"]["krylov_iters_per_ipcs_step"]["pressure_cg"] == 0
    for k in ("device_loop_s3", "device_loop_s1", "host_loop_s1"):    # learning loop: device-resident (S3, S1) and host-driven
        assert res["rates"]["training_loop"][k]["value"] > 0 and res["rates"]["training_loop"][k]["optimiser_steps"] > 0
    for k in ("s1_env_steps_per_s", "s2_ipcs_env_steps_per_s", "training_env_steps_per_s"):
        assert res[k] > 0


def test_bench_gpus_flag_starts_the_ranks_itself(lib_built):
    """`python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset): the parent starts two rank processes, measures the CPU
    baseline itself and relays rank 0's line, which keeps `cpu_baseline` and the rank-local C2 / C3 / C5 entries."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MDQ_SHARE_GPU="1", MDQ_DIST_BACKEND="gloo", MDQ_BENCH_CPU_LEGS="s2only")
    res = _line([sys.executable, "bench.py", "--gpus", "2", "--share-replay", "--cpu-budget", "1"] + SMALL, env)
    assert KEYS | {"cpu_baseline", "collective_backend"} <= set(res)
    assert res["n_gpus"] == 2 and res["collective_backend"]["ranks"] == 2
    assert abs(res["value"] - 2 * 16 * 6 / (res["ms_per_step"] * 6e-3)) < 1e-6 * res["value"]
    assert res["cpu_baseline"]["s2_ipcs"]["value"] > 0
    for k in ("C2_s2_diverged_meshes", "C3_s3_ah93w145", "C5_s2_refined_mesh"):
        assert res["rates"][k].get("value", 0) > 0, (k, res["rates"][k])
    for k in ("device_loop_s3", "device_loop_s1", "host_loop_s1"):   # every loop ran (no fallback), with the record all-gather
        r = res["rates"]["training_loop"][k]
        assert "error" not in r and r["optimiser_steps"] > 0 and r["shared_replay"], (k, r)


def test_bench_gpus_flag_refuses_a_node_with_fewer_gpus(lib_built):
    """One GPU here: `--gpus 2` without the debugging override must fail loudly, not print an n_gpus: 1 line."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with a single GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MDQ_SHARE_GPU")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--no-cpu-baseline"] + SMALL, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "needs 2 GPUs" in out.stderr and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # and a launcher world that disagrees with the flag is refused as well
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--no-cpu-baseline"] + SMALL, cwd=ROOT,
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "disagree" in out.stderr


def test_bench_two_torchrun_ranks_share_one_gpu(lib_built):
    """The driver's form for N > 1: torchrun starts the ranks (headline only here: --s1-steps 0 skips the side measurements)."""
    env = dict(os.environ, MDQ_SHARE_GPU="1", MDQ_DIST_BACKEND="gloo")
    small = list(SMALL)
    small[small.index("--s1-steps") + 1] = "0"
    res = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                 "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2"] + small, env)
    assert res["n_gpus"] == 2 and res["collective_backend"]["ranks"] == 2 and res["config"]["collective_backend"]["ranks"] == 2
    assert abs(res["value"] - 2 * 16 * 6 / (res["ms_per_step"] * 6e-3)) < 1e-6 * res["value"]


def test_bench_one_rank_through_rccl(lib_built):
    """The N > 1 control flow of bench.py over RCCL itself, as far as one GPU allows: MDQ_FORCE_COLLECTIVES=1 makes the single
    rank create an "nccl" process group (communicator on the device) and run its barriers, the max-over-ranks timing
    reductions and the learning loop's gradient all-reduce / record all-gather through it - the calls the driver's
    N = 2, 4, 8 runs make.  (The side measurements are skipped; the learning loop runs.)"""
    small = list(SMALL)
    small[small.index("--s1-steps") + 1] = "0"
    res = _line([sys.executable, "bench.py", "--no-cpu-baseline"] + small, dict(os.environ, MDQ_FORCE_COLLECTIVES="1"))
    assert res["n_gpus"] == 1 and res["collective_backend"]["backend"].startswith("nccl") and res["collective_backend"]["ranks"] == 1
    assert res["value"] > 0 and abs(res["value"] - 16 * 6 / (res["ms_per_step"] * 6e-3)) < 1e-6 * res["value"]


TINY = ["--steps", "3", "--warmup", "1", "--repeats", "2", "--spinup", "20", "--s2-steps", "10", "--envs", "4",
        "--s1-steps", "0", "--train-steps", "0", "--s1-solver-steps", "100", "--no-cpu-baseline"]


def test_bench_eight_ranks_share_one_gpu(lib_built):
    """The launcher at the width the driver uses (N = 8), tiny envs, all ranks on cuda:0 over gloo: eight builds behind the
    rank-0 barrier, eight ground truths, the 127.0.0.1 rendezvous with the short timeout, max-over-ranks timing."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MDQ_SHARE_GPU="1", MDQ_DIST_BACKEND="gloo")
    res = _line([sys.executable, "bench.py", "--gpus", "8"] + TINY, env)
    assert res["n_gpus"] == 8 and res["collective_backend"]["ranks"] == 8 and res["scaling"] == "weak"
    assert abs(res["value"] - 8 * 4 * 3 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]
    assert res["value_min"] <= res["value"] <= res["value_max"]


def _dying_rank_env(tmp_path, rank, **extra):
    """Failure injection that lives in the TESTS (round 4 had a hook inside bench.py's main): a `sitecustomize` on the
    children's PYTHONPATH ends the interpreter whose RANK is `rank` with exit code 3 before its script starts."""
    (tmp_path / "sitecustomize.py").write_text(
        "import os, sys\n"
        f"if os.environ.get('RANK') == '{rank}' and os.environ.get('WORLD_SIZE'):\n"
        f"    sys.stderr.write('rank {rank}: failure injected by the test (sitecustomize), exiting with code 3 before the rendezvous\\n')\n"
        "    sys.stderr.flush()\n"
        "    os._exit(3)\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MDQ_SHARE_GPU="1", MDQ_DIST_BACKEND="gloo", PYTHONPATH=str(tmp_path) + os.pathsep + env.get("PYTHONPATH", ""), **extra)
    return env


def test_bench_launcher_tears_down_when_a_rank_dies(lib_built, tmp_path):
    """Rank 3 of 8 exits at the start: the other seven sit in the rendezvous; the parent names the rank, terminates its
    children and exits non-zero within seconds - not after a c10d timeout."""
    import time
    env = _dying_rank_env(tmp_path, 3)
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8"] + TINY, cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    took = time.monotonic() - t0
    assert out.returncode == 1 and took < 150, (out.returncode, took, out.stderr[-2000:])   # (8 x `import torch` on a cold box)
    assert "rank 3 of 8 exited with code 3" in out.stderr and "failure injected by the test" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
