#!/usr/bin/env python3
"""Benchmark of the MI355X hot path: batched Env2DAirfoil steps on the ys930 mesh.

    python bench.py --gpus N --steps K --warmup W

`value`: a "step" is one batched ENVIRONMENT step of unit-of-work S3 (the north-star step): every
one of the `--envs` (default 128, BASELINE.json configs[1]) environments of a rank removes a vertex and restores the
Delaunay triangulation (Env2DAirfoil.py:452-512), smooths the mesh (flow_solver.py:236-237), interpolates the five
snapshots and evaluates the ten force integrals (Env2DAirfoil.py:547-602, :380-428), builds the state graph
(:244-290), runs ONE IPCS time step on the coarsened mesh (flow_solver.py:362-396, warm start = interpolated last
snapshot) and the Q-network forward (airfoilgcnn.py:85-145); actions follow `default_rng(1370 + ...)` streams with
epsilon = 0.5, so the meshes of the batch diverge; terminated environments are reset in place.  Everything is resident
in HBM before the timed region.  K steps are timed `--repeats` (5) times, each between a barrier + synchronize pair;
value = (ranks x envs x K) / median-over-repeats of the max-over-ranks time (min / max are reported as well).

For N > 1 launch with torchrun (one rank per GPU); environments are independent so they are sharded across ranks
with no data-path collective (weak scaling).

Top-level extras: s1_env_steps_per_s (the reference-semantics step without the flow re-solve), s2_ipcs_env_steps_per_s
(one FlowSolver.evolve per env on 128 identical, developed-flow meshes), s3_env_steps_per_s (= value),
training_env_steps_per_s (S1 rollout + replay + optimiser step, RCCL all-reduce for N > 1).
  roofline      dominant kernel of an S3 step (smooth_kernel): algorithmic bytes per launch / launch duration measured
                with HIP events on the launch stream in this run vs the 8 TB/s HBM peak (a latency-bound kernel: the
                fraction is tiny and says so); `step` = the whole S3 step's algorithmic bytes (S1 part + the IPCS leg in
                the SURVEY 8(d) convention with the measured iteration counts) over the step time
  roofline_s2_velocity   the dominant S2 kernel (at_velocity_kernel), bound by LDS fp64 atomics / FP64 issue: fraction
                by the implemented algorithm's bytes and by the PMC counter traffic (profiles/, rocprofv3 --pmc)
  rates         details of every measurement, incl. BASELINE configs C2 (S2 on diverged meshes), C3 (ah93w145) and C5
                (red-refined mesh)
  cpu_baseline  the numpy/scipy oracle on the host: the same S3 step on one core (bounded sample), plus the IPCS step
                alone on 1 core and on 12 processes (the reference's num_parallel); rank 0 at N = 1 only, measured
                BEFORE the GPU is initialised
"""
import argparse
import os as _os

# single-threaded BLAS / OpenMP pools BEFORE numpy / scipy / torch are imported: the hosts expose hundreds of logical CPUs
# under a small CPU quota, and idle pool threads that keep spinning starve the threads that launch kernels
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    _os.environ.setdefault(_k, "1")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # concurrent streams on separate hardware queues (meshdqn_amd/__init__.py)
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_WARMED = False
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MAX_CLOCK_HZ = 2.4e9   # MI355X_MICROARCH.md: max clock 2400 MHz


def prof_early(name, key):
    """A key of a committed summary under profiles/ (None when absent)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get(key)
    except Exception:  # noqa: BLE001
        return None


def _profiler_attached():
    """The environment test of `meshdqn_amd.streams.profiler_attached` (kept in step with it by tests/test_streams_cpu.py;
    repeated here because the CPU-baseline interpreters do not import torch)."""
    for var in ("LD_PRELOAD", "HSA_TOOLS_LIB"):
        val = os.environ.get(var, "").lower()
        if "rocprof" in val or "roctracer" in val:
            return True
    return any(k.startswith(("ROCPROF_", "ROCP_TOOL", "ROCPROFILER_")) for k in os.environ)


def _quota_note():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return "" if q == "max" else f" under a cgroup quota of {int(q) / int(per):.0f} cores"
    except (OSError, ValueError):
        return ""


def _cpu_worker(args):
    """One oracle environment stepping for `budget_s` seconds (child process of the multi-process CPU baseline)."""
    budget_s, spinup = args
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle.ipcs import OracleFlowSolver
    z = np.load(os.path.join(ROOT, "tests", "golden", "ys930.npz"))
    fs = OracleFlowSolver(z["coords"], z["cells"])
    for _ in range(spinup):
        fs.evolve()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        fs.evolve()
        n += 1
    return n, time.perf_counter() - t0


def _cpu_worker_s3(args):
    """One oracle environment taking S3 env steps for `budget_s` seconds (child of the multi-process CPU baseline)."""
    budget_s, seed = args
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    from oracle.env import OracleEnv
    from oracle.ipcs import OracleFlowSolver
    z = np.load(os.path.join(ROOT, "tests", "golden", "ys930.npz"))
    agent = dict(solver_steps=20, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1,
                 u=-1, p=-1, time_reward=0.005, save_steps=4, goal_vertices=0.95, plot_dir="")
    env = OracleEnv(z["coords"], z["cells"], agent)
    env.get_state()
    rng = np.random.default_rng(1370 + seed)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        env.step(int(rng.integers(0, 180)))
        m = env.flow.mesh
        fs = OracleFlowSolver(m.coords, m.cells, smooth=False)
        fs.u_n, fs.p_n = env.u[-1].copy(), env.p[-1].copy()
        fs.evolve()
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline_parallel_s3(budget_s=10.0, procs=12):
    """BASELINE.md B2 for the headline unit: P independent single-threaded oracle processes taking S3 env steps."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_worker_s3, [(budget_s, k) for k in range(procs)])
    return sum(n / dt for n, dt in res)


def cpu_baseline_parallel(budget_s=8.0, procs=12):
    """The reference's `num_parallel: 12` Ray workers (configs/ray_ys930.yaml:30) mirrored with P independent
    single-threaded oracle processes: aggregate IPCS env-steps/s of the host."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")   # (the parent has initialised the GPU: no fork)
    with ctx.Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(budget_s, 10)] * procs)
    return sum(n / dt for n, dt in res)


def cpu_baseline_s3(budget_s=20.0):
    """The S3 env step of ONE ys930 env with the oracle on one core: OracleEnv.step (scipy Delaunay, smoothing,
    interpolation, probes, state) + FlowSolver on the coarsened mesh (assembly + sparse LU, like the reference's
    remesh under DEPLOY) + one evolve() from the interpolated last snapshot.  Returns (steps, seconds)."""
    from oracle.env import OracleEnv
    from oracle.ipcs import OracleFlowSolver
    z = np.load(os.path.join(ROOT, "tests", "golden", "ys930.npz"))
    agent = dict(solver_steps=20, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1,
                 u=-1, p=-1, time_reward=0.005, save_steps=4, goal_vertices=0.95, plot_dir="")
    env = OracleEnv(z["coords"], z["cells"], agent)
    env.get_state()
    rng = np.random.default_rng(1370)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        env.step(int(rng.integers(0, 180)))
        m = env.flow.mesh
        fs = OracleFlowSolver(m.coords, m.cells, smooth=False)
        fs.u_n, fs.p_n = env.u[-1].copy(), env.p[-1].copy()
        fs.evolve()
        n += 1
    return n, time.perf_counter() - t0


def cpu_baseline(budget_s=15.0, spinup=20):
    """Oracle (kind 'port') on one core: IPCS evolve() steps/s for ONE ys930 env."""
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from oracle.ipcs import OracleFlowSolver
    z = np.load(os.path.join(ROOT, "tests", "golden", "ys930.npz"))
    fs = OracleFlowSolver(z["coords"], z["cells"])
    for _ in range(spinup):
        fs.evolve()
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        fs.evolve()
        n += 1
    dt = time.perf_counter() - t0
    s2 = dict(value=n / dt, unit="IPCS env steps/s", cores=1,
              sample=f"{n} FlowSolver.evolve() steps of 1 ys930 env (numpy/scipy oracle, sparse LU back-substitution like "
                     f"the reference's MUMPS path), {dt:.1f} s on 1 core; mesh smoothing / assembly / factorisation excluded")
    legs = os.environ.get("MDQ_BENCH_CPU_LEGS", "s3,parallel").split(",")   # (debug knob)
    out = dict(value=None, unit="env steps/s", cores=1, kind="port", sample="S3 leg skipped (MDQ_BENCH_CPU_LEGS)", s2_ipcs=s2)
    if "s3" in legs:
        n3, dt3 = cpu_baseline_s3(max(budget_s, 10.0))
        out.update(value=n3 / dt3, sample_short=f"{n3} S3 env steps of 1 ys930 env in {dt3:.1f} s on 1 core (numpy/scipy oracle, kind=port)",
                   sample=f"{n3} S3 env steps of 1 ys930 env in {dt3:.1f} s on 1 core: OracleEnv.step (scipy Delaunay, "
                          "python-loop smoothing / interpolation / probes / state) + Taylor-Hood assembly and sparse LU of "
                          "the coarsened mesh + one evolve() from the interpolated last snapshot (numpy/scipy oracle); no "
                          "Q-network forward (negligible)")
    try:
        procs = 12
        if "parallel" not in legs:
            raise RuntimeError("skipped (MDQ_BENCH_CPU_LEGS)")
        if _profiler_attached():
            raise RuntimeError("skipped under a profiler (its preloaded tool may already hold the GPU: no child interpreters)")
        if "s3" in legs:
            out["parallel_value"] = cpu_baseline_parallel_s3(10.0, procs)
            out["parallel_cores"] = procs
            out["parallel_sample"] = (f"{procs} independent single-threaded oracle processes (BASELINE.md B2, the reference's "
                                      f"num_parallel: 12), 10 s of S3 env steps each, aggregate env-steps/s; host exposes "
                                      f"{os.cpu_count()} logical CPUs" + _quota_note())
        s2["parallel_value"] = cpu_baseline_parallel(8.0, procs)
        s2["parallel_sample"] = (f"{procs} independent single-threaded oracle processes (the reference's num_parallel: 12 "
                                 f"workers), 8 s each, aggregate IPCS env-steps/s; host exposes {os.cpu_count()} logical "
                                 f"CPUs" + _quota_note())
    except Exception as exc:  # noqa: BLE001 - the baseline is informational
        s2["parallel_value"] = None
        s2["parallel_sample"] = f"failed: {exc!r}"
    return out


_REFINED = {}


def _refined_mesh_file(name, levels=1):
    """`<name>_refined`: the lab mesh red-refined once (ys930: 3 322 vertices / 6 280 triangles, BASELINE configs[4]);
    `<name>_refined2`: twice (12 924 / 25 120: the env-step instances of 16 384 vertices) - written once per process to a
    temporary .npz (the environment loads meshes from files)."""
    if (name, levels) not in _REFINED:
        import tempfile
        from meshdqn_amd.ipcs_batch import smooth_coords
        from meshdqn_amd.mesh_ops import red_refine
        from meshdqn_amd.topology import MeshTopology
        z = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
        rc, rcells = red_refine(smooth_coords(MeshTopology(z["coords"], z["cells"]), 50), z["cells"])
        for _ in range(levels - 1):
            rc, rcells = red_refine(rc, rcells)
        fd, path = tempfile.mkstemp(prefix=f"mdq_{name}_refined{levels}_", suffix=".npz")
        os.close(fd)
        np.savez(path, coords=rc, cells=rcells)
        _REFINED[(name, levels)] = path
    return _REFINED[(name, levels)]


def _env_config(args, mesh=None):
    mesh = mesh or args.mesh
    steps = args.s1_solver_steps
    if mesh.endswith("_refined2"):
        path = _refined_mesh_file(mesh[:-len("_refined2")], 2)
        steps = min(steps, 20)
    elif mesh.endswith("_refined"):
        path = _refined_mesh_file(mesh[:-len("_refined")])
        steps = min(steps, 200)        # (ground truth of a rate measurement: 5000 steps of the refined mesh take a minute)
    else:
        path = os.path.join(ROOT, "tests", "golden", f"{mesh}.npz")
    return dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"),
                                 geometry_params=dict(mesh=path),
                                 solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
                agent_params=dict(solver_steps=steps, episodes=10, timesteps=10000, threshold=0.001,
                                  N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1, time_reward=0.005,
                                  save_steps=max(steps // 5, 1), goal_vertices=0.95, plot_dir=""))


_BASE_ENVS = {}


def _timed(dist, dev, fn):
    """fn() between two barrier + synchronize pairs; the max over the ranks of the elapsed time."""
    import torch
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = tt.item()
    return el


def measure_env_steps(args, dev, dist, world, flow_steps=0, steps=None, warmup=8, repeats=1, mesh=None, keep=None, envs=None,
                      flow_pressure="cg"):
    """Units of work S1 / S3 for every env of the batch, all on the GPU: vertex removal + Delaunay
    restoration (mdq_remesh), smooth(50) (mdq_smooth), topology / N-closest / state graph (mdq_env_topology), snapshot
    interpolation + 10 force integrals + node features, fused Q-network forward + epsilon-greedy action; with
    flow_steps > 0 (S3) also the matrix-free IPCS set-up and `flow_steps` IPCS steps on every coarsened mesh.
    `steps` batched steps are timed `repeats` times.  Returns env-steps/s over all ranks (median) and the details."""
    import torch
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.gcn_fused import FusedGcn
    from meshdqn_amd.vec_env import VecEnvGroups
    mesh = mesh or args.mesh
    steps = steps or args.s1_steps
    cfg = _env_config(args, mesh)
    B = envs or args.envs
    if mesh not in _BASE_ENVS:
        _BASE_ENVS[mesh] = Env2DAirfoil(cfg, compute_device=dev)      # ground truth + snapshots: 5000 IPCS steps, once
    base = _BASE_ENVS[mesh]
    G = args.env_groups
    global _WARMED
    if not _WARMED:
        # process-level warm-up: the first rollout of a process runs with 8-20 ms jitter per step for about a second
        # (host thread pool, per-thread heaps, HIP per-thread state); a throw-away rollout absorbs it
        _WARMED = True
        w = VecEnvGroups(cfg, B, G, compute_device=dev, base_env=base, flow_steps=1, flow_rtol=args.rtol)
        wr = np.random.default_rng(0)
        w.rollout(lambda g, env, st: wr.integers(0, 181, env.B), args.s1_warmup)
        del w
    groups = VecEnvGroups(cfg, B, G, compute_device=dev, base_env=base, flow_steps=flow_steps, flow_rtol=args.rtol,
                          flow_overlap=flow_steps > 0 and not args.no_flow_overlap, flow_pressure=flow_pressure)
    torch.manual_seed(0)
    net = NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    net = net.to(dev)
    fused = [FusedGcn(net) for _ in groups.envs]
    calib = []
    if flow_steps > 0 and not args.no_flow_overlap and not args.host_step:
        # a flow stream that really runs beside the group's stream (HIP's stream -> hardware queue mapping: see
        # VecEnv2DAirfoil.calibrate_streams), found by timing a few real steps; the envs are reset afterwards
        for g, env in enumerate(groups.envs):
            with torch.cuda.stream(groups.streams[g]):
                calib.append(env.calibrate_streams(fused[g]))
    rank = int(os.environ.get("RANK", "0"))
    rngs = [np.random.default_rng(1370 + 64 * rank + g) for g in range(len(groups.envs))]

    def act(g, env, st):
        q = fused[g].forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], env.N, env.EMAX)
        greedy = q.argmax(1).cpu().numpy()
        return np.where(rngs[g].random(env.B) < 0.5, rngs[g].integers(0, 181, env.B), greedy)

    def draw(k):
        """The epsilon = 0.5 random streams of `act`, drawn ahead for k steps (same order of draws)."""
        ex, ra = [], []
        for g, env in enumerate(groups.envs):
            pairs = [(rngs[g].random(env.B) < 0.5, rngs[g].integers(0, 181, env.B)) for _ in range(k)]
            ex.append(np.array([p_[0] for p_ in pairs]))
            ra.append(np.array([p_[1] for p_ in pairs]))
        return ex, ra

    if args.host_step:      # state -> host -> action -> host -> step: two host round trips per batched step
        run = lambda k: groups.rollout(act, k)                      # noqa: E731
    else:                   # device-resident: action selection / decoding / reward / reset logic as kernels
        def run(k):
            ex, ra = draw(k)
            groups.rollout_device(fused, k, ex, ra)
    run(warmup)
    if keep is not None and not args.host_step:     # HIP events around every smoothing launch of the timed rollouts
        for e in groups.envs:
            e.smooth_events = []
    times = [_timed(dist, dev, lambda: run(steps)) for _ in range(repeats)]
    el = float(np.median(times))
    venv = groups.envs[0]
    rate = world * B * steps / el
    out = dict(value=rate, unit="env steps/s", ms_per_batched_step=el / steps * 1e3, batched_steps=steps, repeats=repeats,
               value_min=world * B * steps / max(times), value_max=world * B * steps / min(times),
               seconds_per_repeat=times, mesh=mesh, envs_per_gpu=B, env_groups=len(groups.envs),
               vertices_min_max=[int(min(e.nv.min() for e in groups.envs)), int(max(e.nv.max() for e in groups.envs))],
               flow_overlap=bool(flow_steps > 0 and not args.no_flow_overlap), device_resident_step=not args.host_step)
    if flow_steps > 0:
        it = np.concatenate([e.flow_iters.cpu().numpy() for e in groups.envs]).astype(np.float64) / flow_steps
        out["krylov_iters_per_ipcs_step"] = {"velocity_bicgstab": float(it[:, 0].mean()), "pressure_cg": float(it[:, 1].mean()),
                                             "correction_cg": float(it[:, 2].mean())}
    if keep is not None:
        keep.append(groups)
        ev = [p_ for e in groups.envs for p_ in (getattr(e, "smooth_events", None) or [])]
        if ev:
            ms = np.array([a.elapsed_time(b_) for a, b_ in ev])
            out["smooth_kernel_in_rollout_ms"] = dict(mean=float(ms.mean()), min=float(ms.min()), max=float(ms.max()),
                                                      launches=int(ms.size))
            # steady-state period of a batched step: HIP events in front of consecutive smoothing launches on the group's stream,
            # inside the timed rollouts (the first steps of every rollout - pipeline fill - and the rollout boundaries left out)
            ev0 = getattr(groups.envs[0], "smooth_events", None) or []
            per = [ev0[k - 1][0].elapsed_time(ev0[k][0]) for k in range(1, len(ev0)) if (k % steps) >= 5]
            if per:
                out["steady_state_ms_per_step"] = float(np.median(per))
            try:        # the kernel's own diagnostics of the LAST launch of the timed rollouts (per environment)
                from meshdqn_amd.mesh_ops import smooth_fast_stats
                st = smooth_fast_stats(dev, groups.envs[0].B, groups.envs[0].NV, groups.streams[0])
                out["smooth_kernel_in_rollout_ms"]["last_launch_diagnostics"] = dict(
                    environments=int(st.shape[0]), handed_back_to_the_walk=int((st[:, 0] > 0).sum()),
                    sweeps_with_repair_rounds=int(st[:, 1].sum()), repair_rounds=int(st[:, 2].sum()),
                    pipelined_sweeps_redone=int((st[:, 3] & 0xFFFF).sum()), sweeps_in_plain_order=int((st[:, 3] >> 16).sum()))
            except Exception as exc:  # noqa: BLE001
                out["smooth_kernel_in_rollout_ms"]["last_launch_diagnostics"] = repr(exc)
        for e in groups.envs:
            e.smooth_events = None
    return out


def measure_smooth_kernel(dev, groups, reps=20):
    """Launch duration of smooth_kernel (HIP events on the launch stream, this run) on the diverged meshes the S3
    rollout has left in the first env group, all of them smoothed (50 sweeps); algorithmic bytes = coordinates read
    and written + cells read."""
    import torch
    from meshdqn_amd.mesh_ops import smooth_batch_gpu
    dt = groups.envs[0].dtopo
    nv, nt = dt.nv.cpu().numpy(), dt.nt.cpu().numpy()
    its = torch.full_like(dt.nv, 50)
    c0 = dt.coords.clone()
    smooth_batch_gpu(c0.clone(), dt.cells, dt.nv, dt.nt, its)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        c = c0.clone()
        e0.record()
        smooth_batch_gpu(c, dt.cells, dt.nv, dt.nt, its)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    nbytes = float((nv.astype(np.float64) * 32 + nt.astype(np.float64) * 12).sum())
    nblk = (np.maximum(nv - 182, 1) + 31) // 32          # (interior vertices of a ys930-family mesh: all but 182 boundary ones)
    return dict(launch_ms=float(np.mean(ms)), launch_ms_min=float(np.min(ms)), launches=reps, meshes=int(len(nv)),
                block_steps_per_sweep=float(nblk.mean()),          # dependent block steps of one sweep (32-row blocks)
                algorithmic_bytes_per_launch=nbytes,
                workspace_bytes_per_launch=float((nblk * 8192.0 * (1 + 50)).sum()))   # block inverses: written once, read per sweep


def measure_s2_full_chip(args, dev, topo, x, envs=256):
    """The S2 workload with one workgroup on EVERY CU (256 envs per GPU instead of BASELINE's 128, which leave half of
    the chip idle): rate + velocity-kernel roofline, rank-local, short."""
    import torch
    from meshdqn_amd.ipcs_batch import IpcsBatch
    batch = IpcsBatch([topo] * envs, [x] * envs, device=dev, rtol=args.rtol, cell_order=args.cell_order)
    out = (torch.empty((envs, 1), dtype=torch.float64, device=dev), torch.empty((envs, 1), dtype=torch.float64, device=dev))
    for _ in range(args.spinup + 20):
        batch.evolve(1, out=out)
    n = max(args.s2_steps // 2, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        batch.evolve(1, out=out)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    batch.iters.zero_()
    outs = (torch.empty((envs, n), dtype=torch.float64, device=dev), torch.empty((envs, n), dtype=torch.float64, device=dev))
    _, _, kms = batch.evolve_timed(n, out=outs)
    it = batch.iters.cpu().numpy().astype(np.float64) / n
    vb = batch.velocity_kernel_bytes(it)
    return dict(envs_per_gpu=envs, value=envs * n / el, unit="env steps/s (this rank)", ms_per_step=el / n * 1e3,
                velocity_kernel_ms=kms[0] / n, velocity_kernel_roofline_frac=vb / (kms[0] / n * 1e-3) / 1e9 / HBM_PEAK_GBS)


def measure_train(args, dev, dist, world):
    """The learning loop of airfoil_dqn.py:428-503 at scale (BASELINE configs[3]): per batched step every rank steps
    its envs, pushes B transitions, and all ranks take ONE optimiser step on the all-reduced (RCCL) gradient of a
    32-transition minibatch each.  Measured for the device-resident loop (`train_loop_device`: no host round trip inside
    a step) on the S3 env step (the headline's) and on S1, and for the host-driven loop (`train_loop_vec`) on S1.
    Returns env-steps/s over all ranks (value = the S3 device loop)."""
    import torch
    from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_device, train_loop_vec
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    ctx = DistContext(device=dev)                 # re-uses the process group bench.py has initialised
    cfg = _env_config(args)

    def run(loop, flow_steps, n):
        trainer = DQNTrainer(n_actions=180, num_inputs=17, ctx=ctx)
        venv = VecEnv2DAirfoil(cfg, args.envs, compute_device=dev, base_env=_BASE_ENVS[args.mesh], flow_steps=flow_steps,
                               flow_rtol=args.rtol, flow_overlap=flow_steps > 0 and not args.no_flow_overlap)
        kw = dict(share_replay=True) if args.share_replay else {}
        loop(trainer, venv, 4, **kw)              # fills the replay ring past one minibatch, warms every path
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        out = loop(trainer, venv, n, **kw)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = tt.item()
        if hasattr(venv, "flow_wait"):
            venv.flow_wait()
        venv_cal = venv
        del venv
        return dict(value=world * args.envs * n / el, ms_per_batched_step=el / n * 1e3, batched_steps=n,
                    optimiser_steps=len(out["losses"]), last_loss=float(out["losses"][-1]) if out["losses"] else None,
                    stream_calibration_ms=dict(flow=[round(v, 3) for v in getattr(venv_cal, "calibration_ms", [])],
                                               optimiser=[round(v, 3) for v in getattr(trainer, "opt_calibration_ms", [])]),
                    shared_replay=bool(args.share_replay and world > 1),
                    replay_records=int(trainer.device_memory.size()) if trainer.device_memory is not None else None)
    def guarded(*a):       # (a side measurement must not take the bench line down; every rank fails or succeeds alike)
        try:
            return run(*a)
        except Exception as exc:  # noqa: BLE001
            return dict(value=0.0, ms_per_batched_step=None, batched_steps=0, optimiser_steps=0, error=repr(exc))
    s3 = guarded(train_loop_device, 1, args.train_steps)
    s1 = guarded(train_loop_device, 0, args.train_steps)
    host = guarded(train_loop_vec, 0, args.train_steps)
    best = next((r for r in (s3, s1, host) if r["value"] > 0), s3)
    return dict(value=best["value"], unit="env steps/s", ms_per_batched_step=best["ms_per_batched_step"],
                batched_steps=args.train_steps, device_loop_s3=s3, device_loop_s1=s1, host_loop_s1=host,
                what="per batched step: S3 env step for every env (device-resident: Q-forward, epsilon-greedy choice ... reward / "
                     "reset logic as kernels), B transitions into the record ring (mdq_replay_step), one double-DQN optimiser "
                     "step (minibatch 32 per rank: mdq_replay_sample, fused forward of the no-grad network, hand-written "
                     "forward + backward of the trained one = mdq_gcn_train_step, one flat gradient all-reduce over the "
                     "ranks, mdq_adam_step) on a second stream beside the smoothing kernel; host_loop_s1 = the host-driven "
                     "loop (autograd replayed as a HIP graph) on S1 for comparison")


def measure_deploy(args, dev, removals=44):
    """The deployment evaluator (deploy_dqn.py:318-463,495-517) on one ys930 episode at the stock values:
    `removals` scripted removals (default_rng(1370) actions over [0, 180), replayed like the reference's best-episode
    replay), every coarsened mesh + the final mesh re-simulated for `solver_steps` IPCS steps from rest - all of them as ONE
    IpcsBatch (`deploy(batched=True)`); beside it the reference's order (one mesh at a time inside the loop) on the first
    three removals of the same script."""
    from meshdqn_amd.deploy import deploy
    from meshdqn_amd.env import Env2DAirfoil
    base = _BASE_ENVS[args.mesh]

    def env():
        cfg = _env_config(args)
        cfg["agent_params"].update(gt_drag=base.gt_drag, gt_time=base.gt_time, u=base.original_u, p=base.original_p)
        e = Env2DAirfoil(cfg, compute_device=dev)
        e.gt_lift = base.gt_lift
        return e
    script = np.random.default_rng(1370).integers(0, 180, removals).tolist()
    t0 = time.perf_counter()
    out = deploy(env(), actions=script, stop_on_done=False, batched=True)
    t_b = time.perf_counter() - t0
    t0 = time.perf_counter()
    seq = deploy(env(), actions=script[:3], stop_on_done=False, batched=False)
    t_s = time.perf_counter() - t0
    k = 3
    a_, b_ = out["drag_trajectory"][:k + 1], seq["drag_trajectory"]
    rel = float(np.max(np.abs(a_ - b_) / np.maximum(np.abs(b_), 1e-300)))
    return dict(value=t_b, unit="s per deployed episode", removals=removals, solver_steps=int(base.solver_steps),
                resimulated_meshes=int(out["resimulated_meshes"]), rollout_s=out["rollout_seconds"],
                resimulation_s=out["resimulation_seconds"], drag_error_percent=out.get("drag_error_percent"),
                final_vertices=out.get("final_vertices"),
                sequential_3_removals_s=t_s,
                # batched (45 meshes in one IpcsBatch) against the reference's order (one mesh at a time), first rows, all
                # S snapshot drags / lifts of every 5000-step re-simulation: the flow solver's reproducible operator mode
                # makes a mesh's trajectory independent of the run and of the batch it sits in
                max_rel_diff_batched_vs_sequential=rel, tolerance=1e-9, same_first_rows=bool(rel <= 1e-9),
                rows_bitwise_equal=bool(np.array_equal(a_, b_)),
                operator_mode="reproducible (mode -2 -> 2: LDS element tiles, fixed summation order)" if getattr(
                    base.flow_solver, "reproducible", False) else "mode 3 (LDS atomics)",
                rtol=float(base.flow_solver.rtol),
                what="deploy_dqn.py on the HIP path: policy rolled first, then ALL coarsened meshes of the episode and the final "
                     "mesh re-simulated from rest as one batch (one workgroup per mesh); sequential_* = the reference's order "
                     "(re-assemble, re-factorise and re-simulate inside the loop, one mesh on the chip at a time)")


def measure_s2(args, dev, dist, world, topos, xs, steps, warmup, spinup, **kw):
    """Unit of work S2: one FlowSolver.evolve() per env (three launches: velocity, pressure, correction).  Returns the
    rate, the per-kernel durations (HIP events recorded by the library on the launch stream) and the byte counts."""
    import torch
    from meshdqn_amd.ipcs_batch import IpcsBatch
    B = len(topos)
    batch = IpcsBatch(topos, xs, device=dev, rtol=args.rtol, cell_order=args.cell_order, **kw)
    batch.assemble()
    out = (torch.empty((B, 1), dtype=torch.float64, device=dev), torch.empty((B, 1), dtype=torch.float64, device=dev))
    for _ in range(spinup + warmup):
        batch.evolve(1, out=out)
    batch.iters.zero_()

    def run():
        for _ in range(steps):
            batch.evolve(1, out=out)

    elapsed = _timed(dist, dev, run)
    iters = batch.iters.cpu().numpy().astype(np.float64) / steps
    res = dict(value=world * B * steps / elapsed, unit="env steps/s", ms_per_step=elapsed / steps * 1e3, steps=steps,
               envs_per_gpu=B, mode=int(batch.desc.mode), pressure_direct=bool(batch.desc.pd_enabled),
               krylov_iters_per_step={"velocity_bicgstab": float(iters[:, 0].mean()), "pressure": float(iters[:, 1].mean()),
                                      "correction_cg": float(iters[:, 2].mean())},
               drag_env0=float(out[0][0, 0].item()), lift_env0=float(out[1][0, 0].item()))
    return res, batch


def _diverged_meshes(args, n_envs, removals=20):
    """BASELINE C2: `n_envs` ys930 meshes after `removals` scripted interior-vertex removals each (default_rng(1370 +
    env), host engine: Delaunay restoration + smooth(50) per removal) -> list of (MeshTopology, coordinates)."""
    from meshdqn_amd.mesh_ops import remesh_batch
    from meshdqn_amd.topology import MeshTopology
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{args.mesh}.npz"))
    nv0, nt0 = z["coords"].shape[0], z["cells"].shape[0]
    coords = np.repeat(z["coords"][None].astype(np.float64), n_envs, 0).copy()
    cells = np.repeat(np.sort(z["cells"], axis=1).astype(np.int32)[None], n_envs, 0).copy()
    nv, nt = np.full(n_envs, nv0, np.int32), np.full(n_envs, nt0, np.int32)
    assert (remesh_batch(coords, cells, nv, nt, np.full(n_envs, -1, np.int32), 50) == 0).all()
    interior = ~MeshTopology(z["coords"], z["cells"]).on_boundary       # (boundary vertices have the lowest ids and stay)
    nb = int((~interior).sum())
    rngs = [np.random.default_rng(1370 + b) for b in range(n_envs)]
    for _ in range(removals):
        rem = np.array([int(r.integers(nb, nv[b])) for b, r in enumerate(rngs)], np.int32)
        assert (remesh_batch(coords, cells, nv, nt, rem, 50) == 0).all()
    return [(MeshTopology(coords[b, :nv[b]], cells[b, :nt[b]]), coords[b, :nv[b]].copy()) for b in range(n_envs)]


PMC_SUMMARY = "r06_pmc_summary.json"      # profiles/: ONE PMC pass per committed file (tools/refresh_profiles_r06.sh)


def _kernel_sources(kernel):
    """csrc files a kernel (demangled name) is compiled from: its own .hip by namespace + every shared header."""
    ns = kernel.split("::")[0]
    own = {"mdq": ["mdq_ipcs.hip", "mdq_elem.h"], "mdq_smooth_lin": ["mdq_smooth_linear.hip"], "mdq_smoothing": ["mdq_smooth.hip"],
           "mdq_smooth_big": ["mdq_smooth_big.hip"], "mdq_topo": ["mdq_topology.hip"], "mdq_rm": ["mdq_remesh.hip"],
           "mdq_mesh": ["mdq_mesh.hip"], "mdq_gcn": ["mdq_gcn.hip", "mdq_gcn_train.hip"], "mdq_pf": ["mdq_pressure_factor.hip"],
           "mdq_replay": ["mdq_replay.hip"]}.get(ns, [])
    return own + ["mdq_device.h", "mdq_internal.h"]


def pmc_entry(name, leg, kernel):
    """(raw / corrected HBM bytes per launch of `kernel`, source note) from the committed PMC summary - refused PER KERNEL (None +
    the reason) when the summary is missing or was collected from a different version of any source file that kernel is
    compiled from (sha256 of every csrc file stamped at collection; kernel -> files: `_kernel_sources`)."""
    import hashlib
    stamps = prof_early(name, "kernel_sources_sha256") or {}
    ent = ((prof_early(name, "hbm_bytes_per_launch") or {}).get(leg) or {}).get(kernel)
    src = (f"profiles/{name} @ {prof_early(name, 'git_commit') or 'unstamped'} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
           "this command, gfx950 wide-read correction; counters cannot be read inside the run)")
    if not ent or ent.get("corrected") is None:
        return None, src + f" - NO figure for {kernel}"
    for source in _kernel_sources(kernel):
        try:
            now = hashlib.sha256(open(os.path.join(ROOT, "meshdqn_amd", "csrc", source), "rb").read()).hexdigest()
        except OSError:
            now = None
        if stamps.get(source) is None or stamps.get(source) != now:
            return None, src + f" - REFUSED for {kernel}: {source} has changed since the counters were collected (sha256 differs)"
    return ent, src + f"; sha256 of {', '.join(_kernel_sources(kernel))} match"


PMC_OLDER = ("r05_pmc_summary.json",)       # earlier collections: used for a kernel whose sources have not changed since


def pmc_traffic(name, leg, kernel):
    """The newest committed collection that still describes `kernel` (sha256 of its sources): `name`, then PMC_OLDER."""
    ent, src = pmc_entry(name, leg, kernel)
    for older in PMC_OLDER:
        if ent is None:
            ent, src = pmc_entry(older, leg, kernel)
    return (ent["corrected"] if ent else None), src


def _err(exc):
    """A failed side measurement in the line: the exception and where it was raised (last frames)."""
    import traceback
    tb = traceback.extract_tb(exc.__traceback__)
    return dict(error=repr(exc), where=[f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in tb[-4:]])


LINE_LIMIT = 8192        # bytes of the ONE stdout line (the driver reads the tail of stdout; round 5's 22.9 KB line was not parsed)
DETAIL_FILE = "bench_detail.json"


def _kernel_scratch():
    """Per-kernel resources of the library that RAN (the table is stamped with the sha256 of the .so it was built with;
    refused for any other library)."""
    try:
        from meshdqn_amd import build as _b
        from meshdqn_amd._lib import LIB_PATH
        kres = _b.library_resources(LIB_PATH)
        hot = ("smooth_linear", "at_velocity", "at_pressure", "at_correction", "topology_kernel", "gcn_embed", "mlp_head_c128",
               "remesh_kernel", "evolve_mf", "evolve_kernel<5", "evolve_team_tiles", "setup_matfree", "interpolate", "env_finish",
               "probe")
        return dict(kernels=len(kres), with_scratch={k: v["scratch_bytes_per_lane"] for k, v in kres.items() if v["scratch_bytes_per_lane"] > 0},
                    hot_kernels={k: dict(vgprs=v["vgprs"], scratch=v["scratch_bytes_per_lane"], occupancy=v["occupancy"])
                                 for k, v in kres.items() if any(t in k for t in hot)},
                    unit="bytes per lane (-Rpass-analysis=kernel-resource-usage of this build)")
    except Exception as exc:  # noqa: BLE001
        return _err(exc)


def _write_detail(res):
    """Everything measured (prose, per-repeat times, side tables) goes to a FILE: MDQ_BENCH_DETAIL, default bench_detail.json
    beside this script (a temporary file when that directory is read-only).  Returns the path written."""
    import tempfile
    path = os.environ.get("MDQ_BENCH_DETAIL") or os.path.join(ROOT, DETAIL_FILE)
    try:
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
    except OSError:
        fd, path = tempfile.mkstemp(prefix="mdq_bench_detail_", suffix=".json")
        with os.fdopen(fd, "w") as f:
            json.dump(res, f, indent=1)
    return path


def _num(v, digits=6):
    """A float with `digits` significant digits (the line carries numbers, the detail file carries them in full)."""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    return float(f"{v:.{digits}g}")


def _short(v, n=120):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 3] + "..."


def compact_line(res, detail_path):
    """The contract line from the full result: contract keys only, one NUMBER per side rate, no prose (strings <= 120
    characters), <= LINE_LIMIT bytes.  Everything else is in the detail file."""
    cfg, roof, cpu = res["config"], res["roofline"], res.get("cpu_baseline")
    line = {k: _num(res[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "steady_state_ms_per_step",
                                       "repeats", "value_min", "value_max", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": cfg["workload_short"], "mesh": cfg["mesh"], "envs_per_gpu": cfg["envs_per_gpu"],
                      "rtol": cfg["rtol"], "dt": cfg["dt"], "mu": cfg["mu"], "rho": cfg["rho"],
                      "krylov_iters_per_ipcs_step": {k: _num(v, 4) for k, v in cfg["krylov_iters_per_ipcs_step"].items()},
                      "parallelism": cfg["parallelism"], "collective_backend": cfg["collective_backend"]}
    line["roofline"] = {k: _num(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    line["roofline"].update(kernel="smooth_linear_kernel", traffic_source=_short(roof.get("traffic_source_short")),
                            **{k: _num(roof.get(k)) for k in ("launch_ms", "launches_timed", "launch_ms_alone", "algorithmic_bytes_per_launch",
                                                               "share_of_step", "floor_ms", "launch_over_floor", "device_copy_GBs_same_run")})
    if cpu is not None:
        s2 = cpu.get("s2_ipcs") or {}
        line["cpu_baseline"] = {"value": _num(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                                "sample": _short(cpu.get("sample_short") or cpu.get("sample")),
                                "parallel_value": _num(cpu.get("parallel_value")), "parallel_cores": cpu.get("parallel_cores"),
                                "s2_ipcs": {"value": _num(s2.get("value")), "cores": s2.get("cores"),
                                            "parallel_value": _num(s2.get("parallel_value")), "parallel_cores": 12 if s2.get("parallel_value") else None}}
    rates = {}
    for name, r in (res.get("rates") or {}).items():
        if isinstance(r, dict):
            rates[name] = _num(r["value"]) if isinstance(r.get("value"), (int, float)) else ("error" if "error" in r else None)
        else:
            rates[name] = None
    line["rates"] = rates
    line["rates_unit"] = "env steps/s (deploy_episode_s: seconds per deployed episode)"
    c5 = (res.get("rates") or {}).get("C5_s3_refined_mesh") or {}
    if isinstance(c5.get("krylov_iters_per_ipcs_step"), dict):
        line["c5_s3_krylov_iters"] = {k: _num(v, 4) for k, v in c5["krylov_iters_per_ipcs_step"].items()}
    r5 = ((res.get("rates") or {}).get("C5_s2_refined_mesh") or {}).get("roofline")
    if isinstance(r5, dict):        # the genuinely HBM-bound kernel of the path (BASELINE configs[4]): its own roofline numbers
        line["roofline_c5_s2"] = {k: _num(r5.get(k)) for k in ("bound", "kernel", "launch_ms", "algorithmic_bytes_per_launch", "achieved",
                                                              "peak", "unit", "frac", "traffic", "traffic_raw")}
    line["detail"] = os.path.basename(detail_path)
    return line


def _detail_digest(res):
    """A few human-readable lines for stderr."""
    out = [f"[bench] {res['value']:.0f} {res['unit']} ({res['ms_per_step']:.4f} ms per batched step, steady state "
           f"{res.get('steady_state_ms_per_step')}), {res['n_gpus']} GPU(s)"]
    for name, r in (res.get("rates") or {}).items():
        if isinstance(r, dict):
            out.append(f"[bench]   {name}: {r.get('value')} {r.get('unit', '')}" + (f"  ERROR {r['error']}" if "error" in r else ""))
    return "\n".join(out)


def _stream_log():
    try:
        from meshdqn_amd import streams
        return list(streams.LOG)
    except Exception:  # noqa: BLE001
        return None


def _visible_gpus():
    return _launcher().visible_gpus()


# the rank launcher lives in the package (train.py --gpus N uses it too); imported lazily by file path so that this parent
# process never runs the package's __init__ chain
def _launcher():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mdq_launcher", os.path.join(ROOT, "meshdqn_amd", "launcher.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_watched(cmds, timeout_s, poll=0.2):
    return _launcher().run_watched(cmds, timeout_s, poll)


def launch_ranks(args, argv):
    """`bench.py --gpus N` started by hand or by the driver as ONE process: this parent - which never imports torch or
    touches a GPU - measures the CPU baseline (child interpreters), then starts N fresh rank processes (one per GPU:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment, the same command line), relays
    rank 0's JSON line, and WATCHES ALL OF THEM: the first rank that exits non-zero (or the launch timeout,
    MDQ_LAUNCH_TIMEOUT seconds, default 1700) tears the job down within seconds - the remaining children are terminated,
    the failed rank and its last output lines are named on stderr, the exit code is non-zero and no result line is
    printed (the reference's Ray trainer ends the job when a worker dies as well: airfoil_dqn.py:508-514).  Output of the
    ranks is relayed line by line with a `[rank r]` prefix."""
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    have = _visible_gpus()
    if have is not None and have < n and not os.environ.get("MDQ_SHARE_GPU"):
        sys.stderr.write(f"[bench] --gpus {n} needs {n} GPUs on this node, the driver exposes {have}: refusing "
                         "(MDQ_SHARE_GPU=1 MDQ_DIST_BACKEND=gloo lets several ranks share a GPU, for debugging only)\n")
        return 2
    env = dict(os.environ)
    cpu_file = None
    if not args.no_cpu_baseline:
        try:
            txt = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-budget",
                                  str(args.cpu_budget)], check=True, capture_output=True, text=True).stdout
            fd, cpu_file = tempfile.mkstemp(prefix="mdq_cpu_", suffix=".json")
            with os.fdopen(fd, "w") as f:
                f.write(txt.strip().splitlines()[-1])
            env["MDQ_BENCH_CPU_JSON"] = cpu_file
        except Exception as exc:  # noqa: BLE001 - the baseline is informational
            sys.stderr.write(f"[bench] cpu baseline child failed ({exc!r})\n")
    rc, out0 = _launcher().start_ranks(os.path.abspath(__file__), argv, n, env=env, tag="bench")
    if cpu_file:
        os.unlink(cpu_file)
    if rc != 0:
        return rc
    lines = [l for l in out0 if l.startswith("{")]
    if not lines:
        sys.stderr.write("[bench] rank 0 printed no JSON line\n")
        return 1
    print(lines[-1].rstrip("\n"), flush=True)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="K: batched S3 env steps per timed repeat")
    ap.add_argument("--warmup", type=int, default=10, help="W: untimed batched S3 env steps in front")
    ap.add_argument("--repeats", type=int, default=5, help="timed repeats of the K steps (median reported)")
    ap.add_argument("--envs", type=int, default=128, help="environments per GPU")
    ap.add_argument("--spinup", type=int, default=300, help="S2: untimed IPCS steps from rest before warmup")
    ap.add_argument("--s2-steps", type=int, default=200)
    ap.add_argument("--mesh", default="ys930")
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--s1-steps", type=int, default=50, help="batched steps of the S1 / C3 side measurements (0 = skip all side measurements)")
    ap.add_argument("--cell-order", default="auto", choices=["auto", "mesh", "conflictfree", "morton"])
    ap.add_argument("--train-steps", type=int, default=20, help="batched learning-loop steps (0 = skip)")
    ap.add_argument("--share-replay", action="store_true",
                    help="learning loops with the shared replay: every rank all-gathers the transition records of all ranks")
    ap.add_argument("--s1-warmup", type=int, default=40)
    ap.add_argument("--env-groups", type=int, default=1, help="concurrently stepped env groups per GPU for S1 / S3")
    ap.add_argument("--s1-solver-steps", type=int, default=5000, help="IPCS steps of the ground-truth reset()")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2 / C3 / C5 side measurements")
    ap.add_argument("--host-step", action="store_true",
                    help="S1 / S3: the step() path with its two host round trips per batched step instead of rollout_device")
    ap.add_argument("--no-flow-overlap", action="store_true",
                    help="S3: run the IPCS step of an env step in line instead of beside the next step's mesh kernels")
    args = ap.parse_args(argv)

    if args.cpu_baseline_child:                      # child interpreter: CPU legs only, JSON on stdout
        print(json.dumps(cpu_baseline(args.cpu_budget)), flush=True)
        return 0
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args, argv)              # N fresh rank processes; this one stays off the GPU
    if int(env_world or "1") != args.gpus:
        sys.stderr.write(f"[bench] --gpus {args.gpus} but WORLD_SIZE={env_world}: the launcher's world and the flag disagree, "
                         "refusing to report a line with the wrong n_gpus\n")
        return 2

    # CPU baseline FIRST, before anything initialises the GPU: its multi-process leg starts child interpreters
    # (fork + exec), which must not happen from a process that already holds the device
    cpu = None
    if os.environ.get("MDQ_BENCH_CPU_JSON") and int(os.environ.get("RANK", "0")) == 0:
        try:                                         # measured by the launching parent (launch_ranks) before the ranks started
            cpu = json.load(open(os.environ["MDQ_BENCH_CPU_JSON"]))
        except Exception as exc:  # noqa: BLE001
            sys.stderr.write(f"[bench] cannot read the parent's cpu baseline ({exc!r})\n")
    elif int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        # in a child interpreter (started before this process touches the GPU): whatever the numpy / scipy / Qhull
        # heavy oracle run leaves behind in the process state was measured to slow the later learning loop by 30 %
        import subprocess
        try:
            txt = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-budget",
                                  str(args.cpu_budget)], check=True, capture_output=True, text=True).stdout
            cpu = json.loads(txt.strip().splitlines()[-1])
        except Exception as exc:  # noqa: BLE001 - fall back to the in-process measurement
            sys.stderr.write(f"[bench] cpu baseline child failed ({exc!r}); measuring in-process\n")
            cpu = cpu_baseline(args.cpu_budget)

    # the contract is ONE JSON line on stdout: whatever the measured code prints (the environment keeps the reference's
    # progress messages, e.g. "MAXIMUM REMOVALS REACHED") goes to stderr
    # ... at the level of the file descriptor: RCCL prints its version banner to the C stdout when a communicator is created
    # (lazily: at the first collective - which may be the LAST barrier of this program), and the driver reads the tail of stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    import torch
    # the GPU boxes expose 256 logical CPUs under a 16-core quota: a 256-thread intra-op pool that keeps spinning after
    # any stray CPU tensor op starves the threads that matter (measured: 25 ms per autograd backward)
    torch.set_num_threads(1)   # (the CPU-side tensor ops of this program are tiny: no intra-op pool at all)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rccl = None
    # MDQ_FORCE_COLLECTIVES=1: a job of ONE rank creates its process group too and runs its barriers, timing reductions and
    # the learning loop's collectives through the backend (RCCL) - what a single-GPU box can exercise of the N > 1 path
    forced = world == 1 and os.environ.get("MDQ_FORCE_COLLECTIVES", "") == "1"
    if world > 1 or forced:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced and not os.environ.get("MASTER_PORT"):
            import socket
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            s_.close()
        # MDQ_SHARE_GPU=1 + MDQ_DIST_BACKEND=gloo: debugging aid that lets the multi-rank control flow be exercised on
        # a box with fewer GPUs than ranks (RCCL refuses two ranks on one device); never set by the driver
        ndev = torch.cuda.device_count()
        if ndev < 1 or (local_rank >= ndev and not os.environ.get("MDQ_SHARE_GPU")):
            sys.stderr.write(f"[bench] rank {rank}: local rank {local_rank} has no GPU of its own ({ndev} visible): --gpus "
                             f"{args.gpus} needs {args.gpus} GPUs on this node\n")
            return 2
        dev_index = local_rank % ndev if os.environ.get("MDQ_SHARE_GPU") else local_rank
        torch.cuda.set_device(dev_index)
        backend = os.environ.get("MDQ_DIST_BACKEND", "nccl")
        # rendezvous with a SHORT timeout (a rank that never shows up fails the job in MDQ_RENDEZVOUS_TIMEOUT seconds, not in
        # c10d's 10-30 minutes); the collectives keep a long one: ranks >= 1 wait at the final barrier while rank 0 runs its
        # rank-local side measurements
        import datetime
        pg_kw = dict(timeout=datetime.timedelta(seconds=float(os.environ.get("MDQ_DIST_TIMEOUT", "1800"))))
        store = None
        if "TORCHELASTIC_RUN_ID" not in os.environ:      # (under torchrun the agent owns the store and watches its workers)
            store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, rank == 0,
                                  timeout=datetime.timedelta(seconds=float(os.environ.get("MDQ_RENDEZVOUS_TIMEOUT", "180"))))
            pg_kw.update(store=store, rank=rank, world_size=world)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), **pg_kw)
            try:
                rccl = dict(backend="nccl (RCCL)", version=".".join(str(v) for v in torch.cuda.nccl.version()), ranks=world)
            except Exception:  # noqa: BLE001
                rccl = dict(backend="nccl (RCCL)", ranks=world)
        else:
            dist.init_process_group(backend, **pg_kw)
            rccl = dict(backend=backend, ranks=world)
        if store is not None:       # (the short timeout was for the rendezvous: c10d does not re-time a store it is handed)
            store.set_timeout(pg_kw["timeout"])
    else:
        dist = None
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    from meshdqn_amd import build as _b
    if rank == 0:
        _b.build()
    if dist is not None:
        dist.barrier()
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.topology import MeshTopology
    import gc
    side = args.s1_steps > 0

    # ---------------- headline: S3, K steps x repeats
    gc.collect()
    gc.disable()            # no collector pause inside the timed regions
    keep = []
    s3 = measure_env_steps(args, dev, dist, world, 1, steps=args.steps, warmup=args.warmup, repeats=args.repeats, keep=keep)
    smk = measure_smooth_kernel(dev, keep[0])
    del keep
    gc.enable()

    # ---------------- S2 on the BASELINE configuration (128 identical meshes, developed flow)
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{args.mesh}.npz"))
    topo = MeshTopology(z["coords"], z["cells"])
    x = smooth_coords(topo, 50)
    B = args.envs
    s2, batch = measure_s2(args, dev, dist, world, [topo] * B, [x] * B, args.s2_steps, 20, args.spinup)
    # second pass over the same number of steps with HIP events around every kernel launch (recorded by the
    # library on the launch stream): average duration of each of the three kernels of a step
    batch.iters.zero_()
    n2 = args.s2_steps
    _, _, kms = batch.evolve_timed(n2, out=(torch.empty((B, n2), dtype=torch.float64, device=dev),
                                           torch.empty((B, n2), dtype=torch.float64, device=dev)))
    iters2 = batch.iters.cpu().numpy().astype(np.float64) / n2
    k_vel, k_prs, k_cor = (m / n2 for m in kms)
    vel_bytes = batch.velocity_kernel_bytes(iters2)
    vel_flops = batch.velocity_kernel_flops(iters2)
    survey_bytes = batch.algorithmic_bytes_per_step(iters2)  # SURVEY 8(d) assembled-CSR convention, whole step
    step_bytes = batch.implemented_bytes_per_step(iters2)     # bytes the implemented algorithm moves, whole step
    # the IPCS leg of an S3 step in the SURVEY 8(d) convention, with the iteration counts measured in the S3 rollout
    it3 = s3["krylov_iters_per_ipcs_step"]
    s3_ipcs_bytes = batch.algorithmic_bytes_per_step(np.tile(np.array([[it3["velocity_bicgstab"], it3["pressure_cg"],
                                                                         it3["correction_cg"]]]), (B, 1)))
    del batch

    # device-copy microbenchmark in the same run (check the 8 TB/s spec figure used as `peak`)
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
    dst = torch.empty_like(src)
    dst.copy_(src)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(10):
        dst.copy_(src)
    c1.record()
    torch.cuda.synchronize()
    copy_gbs = 10 * 2 * src.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9   # read + write
    del src, dst

    full = measure_s2_full_chip(args, dev, topo, x) if side else None
    s1 = measure_env_steps(args, dev, dist, world, 0, repeats=3) if side else None
    tr = measure_train(args, dev, dist, world) if side and args.train_steps > 0 else None
    cfgs = {}
    if side and not args.no_configs and rank == 0:
        # BASELINE configs beyond C1 / C2-identical: iteration counts included, RANK-LOCAL (rank 0's GPU; the other ranks
        # wait at the final barrier), short
        try:
            dm = _diverged_meshes(args, B)
            c2, b2 = measure_s2(args, dev, None, 1, [t for t, _ in dm], [c for _, c in dm], 50, 10, args.spinup,
                                pressure_direct="device")
            # the factorisation itself (once per mesh, like the reference's MUMPS factorisation at remesh)
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record()
            for _ in range(5):
                b2.factorize_pressure_device()
            f1.record()
            torch.cuda.synchronize()
            c2["what"] = ("C2: S2 on 128 DIFFERENT ys930 meshes (20 scripted removals each, default_rng(1370 + env), Delaunay "
                          "restoration + smooth(50) per removal), developed flow, direct pressure solve on factors built ON THE "
                          "DEVICE for every coarsened mesh (mdq_ipcs_factorize_pressure, once per mesh, outside the timed steps)")
            c2["factorisation_ms_per_batch"] = f0.elapsed_time(f1) / 5
            c2["factorisation_status_ok"] = bool((b2.pd_status == 0).all().item())
            c2["vertices_min_max"] = [int(min(t.nv for t, _ in dm)), int(max(t.nv for t, _ in dm))]
            del b2
            cg, b2 = measure_s2(args, dev, None, 1, [t for t, _ in dm], [c for _, c in dm], 50, 10, args.spinup,
                                pressure_direct=False)
            del b2
            c2["jacobi_cg_variant"] = dict(value=cg["value"], ms_per_step=cg["ms_per_step"],
                                           krylov_iters_per_step=cg["krylov_iters_per_step"])
            cfgs["C2_s2_diverged_meshes"] = c2
        except Exception as exc:  # noqa: BLE001 - side measurement
            cfgs["C2_s2_diverged_meshes"] = _err(exc)
        try:
            c3 = measure_env_steps(args, dev, None, 1, 1, steps=args.s1_steps, repeats=3, mesh="ah93w145")
            c3["what"] = "C3: the S3 step on the second airfoil geometry (ah93w145, 797 vertices), 128 envs"
            cfgs["C3_s3_ah93w145"] = c3
        except Exception as exc:  # noqa: BLE001
            cfgs["C3_s3_ah93w145"] = _err(exc)
        try:
            sd = measure_env_steps(args, dev, None, 1, 1, steps=args.s1_steps, repeats=3, flow_pressure="direct")
            sd["what"] = ("the S3 step with the pressure matrix of every coarsened mesh re-factorised on the device in every step "
                          "(mdq_ipcs_factorize_pressure: what the reference's MUMPS does at a remesh) and a direct pressure solve: 0 "
                          "Krylov iterations in the pressure solve; the flow leg grows from 1.05 to 1.7 ms on its stream and still "
                          "mostly hides beside the smoothing kernel (VecEnv2DAirfoil(flow_pressure='direct')); the headline keeps "
                          "the Jacobi-CG: one solve per mesh does not pay for a factorisation")
            cfgs["S3_refactorised_pressure"] = sd
        except Exception as exc:  # noqa: BLE001
            cfgs["S3_refactorised_pressure"] = _err(exc)
        try:
            f3 = measure_env_steps(args, dev, None, 1, 1, steps=args.s1_steps, repeats=3, envs=2 * B)
            f3["what"] = (f"the S3 step with {2 * B} envs on the GPU (one workgroup per env on every CU): at the BASELINE batch of "
                          f"{B} the kernels of the step are latency-bound with half of the chip idle; twice the batch costs "
                          "less than twice the time (the IPCS leg no longer hides beside a smoothing kernel that owns every CU)")
            cfgs["S3_full_chip"] = f3
        except Exception as exc:  # noqa: BLE001
            cfgs["S3_full_chip"] = _err(exc)
        try:
            r1 = measure_env_steps(args, dev, None, 1, 0, steps=max(args.s1_steps // 5, 2), warmup=2, repeats=3, mesh=f"{args.mesh}_refined")
            r1["what"] = ("the S1 env step (vertex removal + Delaunay restoration + smooth(50) + interpolation + forces + state graph + "
                          "Q-forward, device-resident) on the red-refined mesh (BASELINE configs[4]: 3 322 vertices / 6 280 triangles): the "
                          "large-mesh instances of the removal / topology kernels (tables on a slab in global memory) and the "
                          "level-scheduled smoothing kernel - the env step of rounds 1-3 refused meshes above 1024 vertices; ground truth "
                          "of 200 solver steps (a rate measurement)")
            cfgs["C5_s1_refined_mesh"] = r1
        except Exception as exc:  # noqa: BLE001
            cfgs["C5_s1_refined_mesh"] = _err(exc)
        try:
            r3 = measure_env_steps(args, dev, None, 1, 1, steps=max(args.s1_steps // 10, 2), warmup=2, repeats=3, mesh=f"{args.mesh}_refined")
            r3["what"] = ("the S3 (north-star) env step on the red-refined mesh: the S1 step above + one IPCS step on every coarsened "
                          "mesh on the flow stream - index data from the large-mesh topology instance, mdq_ipcs_setup_matfree, the "
                          "element operators with global vectors and two workgroups per environment (mode 7 through the dof <- slot lists), "
                          "Jacobi-CG pressure solve")
            cfgs["C5_s3_refined_mesh"] = r3
        except Exception as exc:  # noqa: BLE001
            cfgs["C5_s3_refined_mesh"] = _err(exc)
        try:
            rb = measure_env_steps(args, dev, None, 1, 0, steps=2, warmup=1, repeats=3, mesh=f"{args.mesh}_refined2", envs=min(B, 32))
            rb["what"] = ("the S1 env step on the TWICE red-refined mesh (12 924 vertices / 25 120 triangles: the 16 384-vertex instances "
                          "of the removal / smoothing / topology kernels, every table on the caller's slab - coverage, not speed), "
                          f"{min(B, 32)} environments")
            cfgs["C5b_s1_twice_refined"] = rb
        except Exception as exc:  # noqa: BLE001
            cfgs["C5b_s1_twice_refined"] = _err(exc)
        try:
            cfgs["deploy_episode_s"] = measure_deploy(args, dev)
        except Exception as exc:  # noqa: BLE001
            cfgs["deploy_episode_s"] = _err(exc)
        try:
            from meshdqn_amd.mesh_ops import red_refine
            rc_, rcells = red_refine(x, z["cells"])
            rt = MeshTopology(rc_, rcells)
            c5, b5 = measure_s2(args, dev, None, 1, [rt] * B, [rc_] * B, 20, 5, 100)
            it5 = np.tile(np.array([[c5["krylov_iters_per_step"][k] for k in ("velocity_bicgstab", "pressure", "correction_cg")]]), (B, 1))
            by5 = b5.algorithmic_bytes_per_step(it5)
            c5_pmc, c5_pmc_src = pmc_entry(PMC_SUMMARY, "c5", "mdq::evolve_team_tiles_kernel<false>")
            alg5 = b5.tile_mode_bytes_per_step(it5)       # algorithmic bytes of the element-tile kernel (one launch = one step)
            c5.update(what="C5: S2 on ys930 red-refined once (the mesh does not fit the LDS-resident modes: auto mode takes the element "
                           "tiles with GLOBAL vectors and - while two workgroups per environment fit the chip - TWO workgroups per "
                           "environment, mode 7: bitwise reproducible run to run; mode 5, one workgroup, ran here until round 5: "
                           "11.7 k env-steps/s; round 3 ran the assembled SELL operators, 6.65 k), 100 spin-up steps from rest; "
                           "direct pressure solve from host-built factors", vertices=int(rt.nv), triangles=int(rt.nt),
                      survey_csr_bytes_per_step=by5, survey_equivalent_GBs=by5 / (c5["ms_per_step"] * 1e-3) / 1e9,
                      survey_equivalent_over_hbm_peak=by5 / (c5["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      measured_hbm_bytes_per_launch=c5_pmc, measured_hbm_source=c5_pmc_src,
                      # the HBM-BOUND kernel of the path (evolve_team_tiles_kernel: one launch = one IPCS step of the batch)
                      roofline=dict(bound="hbm", kernel="evolve_team_tiles_kernel<false>", launch_ms=c5["ms_per_step"],
                                    algorithmic_bytes_per_launch=alg5, achieved=alg5 / (c5["ms_per_step"] * 1e-3) / 1e9,
                                    peak=HBM_PEAK_GBS, unit="GB/s", frac=alg5 / (c5["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    traffic=None if not c5_pmc else c5_pmc["corrected"],
                                    traffic_raw=None if not c5_pmc else c5_pmc["raw"],
                                    note="launch_ms = the step time (three small launches beside it: < 1 %); algorithmic bytes: "
                                         "IpcsBatch.tile_mode_bytes_per_step (no cache credit)"),
                      note="survey_* = the bytes an assembled-CSR implementation would stream (SURVEY 8(d) convention) over the step "
                           "time - NOT a roofline fraction of this kernel; measured_hbm_bytes_per_launch = rocprofv3 FETCH_SIZE + "
                           "WRITE_SIZE of the same kernel (profiles/" + PMC_SUMMARY + "; one launch = one step of 128 environments)")
            del b5
            cfgs["C5_s2_refined_mesh"] = c5
        except Exception as exc:  # noqa: BLE001
            cfgs["C5_s2_refined_mesh"] = _err(exc)

    if rank == 0:
        def prof(name, key):
            fp = os.path.join(ROOT, "profiles", name)
            try:
                return json.load(open(fp)).get(key)
            except Exception:  # noqa: BLE001
                return None

        sm_traffic, sm_traffic_src = pmc_traffic(PMC_SUMMARY, "s3", "mdq_smooth_lin::smooth_linear_kernel")
        nt, nv = topo.nt, topo.nv
        insitu = s3.get("smooth_kernel_in_rollout_ms")
        sm_ms = insitu["mean"] if insitu else smk["launch_ms"]      # the launches of the timed region themselves
        sm_gbs = smk["algorithmic_bytes_per_launch"] / (sm_ms * 1e-3) / 1e9
        s3_step_bytes = (0.7e6 * B + s3_ipcs_bytes)            # S1 part 0.7 MB per env step + the IPCS leg
        s3_gbs = s3_step_bytes / (s3["ms_per_batched_step"] * 1e-3) / 1e9
        vel_gbs = vel_bytes / (k_vel * 1e-3) / 1e9
        vel_traffic = prof("traffic.json", "velocity_kernel_hbm_bytes_per_launch")
        res = {
            "metric": "env steps/sec (ys930 ~2k-tri)",
            "value": s3["value"],
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": s3["ms_per_batched_step"],
            # period of a batched step inside the rollouts (HIP events; ms_per_step also pays the fill and the drain of the
            # two-stream pipeline and the one read-back per rollout: K = --steps batched steps per rollout)
            "steady_state_ms_per_step": s3.get("steady_state_ms_per_step"),
            "repeats": args.repeats,
            "value_min": s3["value_min"],
            "value_max": s3["value_max"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "s3_env_steps_per_s": s3["value"],
            "s1_env_steps_per_s": None if s1 is None else s1["value"],
            "s2_ipcs_env_steps_per_s": s2["value"],
            "training_env_steps_per_s": None if tr is None else tr["value"],
            "collective_backend": rccl,          # backend, RCCL version, ranks (None for a single rank)
            "stream_setup": _stream_log(),       # rank 0: how the main / flow / optimiser streams were chosen
            "config": {
                "workload_short": f"{args.mesh} ({nv} vertices / {nt} triangles), {B} envs per GPU, S3 env step (remesh+smooth+interp+"
                                  f"forces+state+1 IPCS step+Q-fwd)",
                "mesh": args.mesh,
                "workload": f"{args.mesh} ({nv} vertices / {nt} triangles), {B} batched envs per GPU, step = S3 (north-star env "
                            f"step): remove vertex + Delaunay restoration + smooth(50) + 5-snapshot interpolation + 10 force "
                            f"integrals + state graph + ONE IPCS step on every coarsened mesh (matrix-free, Jacobi-BiCGStab / "
                            f"CG, rtol {args.rtol:g}, warm start = interpolated last snapshot"
                            + ("" if args.no_flow_overlap else "; it runs on a second stream beside the NEXT step's removal / "
                               "smoothing / topology kernels, its drag / lift are delivered one step later")
                            + ") + fused Q-network forward"
                            + ("" if args.host_step else "; action selection / decoding, reward / terminal logic and in-place "
                               "resets are kernels too (rollout_device: no host round trip inside a step)") + "; "
                            f"epsilon = 0.5 actions from default_rng(1370 + ...), meshes diverge (vertices "
                            f"{s3['vertices_min_max'][0]}..{s3['vertices_min_max'][1]} at the end), terminated envs reset in place",
                "envs_per_gpu": B, "rtol": args.rtol, "dt": 1e-3, "mu": 1e-3, "rho": 1.0,
                "numerics": {
                    "ipcs": "mode 3: operator results accumulated with LDS fp64 atomics => reproducible to round-off only; run-to-run "
                            "spread of drag / lift after 5000 steps 1e-8 .. 1.6e-6 relative (tools/traj_spread.py), inside the 1e-4 "
                            "contract (5000-step values within 1e-5 of the FEniCS CSV rows, tests/test_golden_gpu.py); the correction "
                            "solve starts fused on 15 of 16 steps and takes the |b| of its stopping test from the last exact start",
                    "bitwise_reproducible": "vertex removal / Delaunay restoration, smoothing, topology / N-closest / state graph, "
                                            "fused Q-forward, learning-step gradient (per-graph slices reduced in graph order)"},
                "krylov_iters_per_ipcs_step": s3["krylov_iters_per_ipcs_step"],
                "parallelism": f"dp{world} (independent envs sharded, no data-path collective)",
                "collective_backend": rccl,
                "reference_published": "45.8 IPCS steps/s for 1 env (FEniCS, unknown hardware; BASELINE.md) - context only",
            },
            "roofline": {"bound": "hbm", "achieved": sm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sm_gbs / HBM_PEAK_GBS,
                         "traffic": sm_traffic,
                         "kernel": "smooth_linear_kernel (DOLFIN smooth(50) as blocked triangular solves: 32-row block inverses built "
                                   "per launch, 22 dependent block steps per sweep, validation in parallel, limited steps repaired)",
                         "launch_ms": sm_ms, "launches_timed": insitu["launches"] if insitu else smk["launches"],
                         "launch_ms_min_max": [insitu["min"], insitu["max"]] if insitu else None,
                         "launch_ms_alone": smk["launch_ms"],
                         "last_launch_diagnostics": insitu.get("last_launch_diagnostics") if insitu else None,
                         "traffic_source": sm_traffic_src,
                         "traffic_source_short": (sm_traffic_src.split(" @ ")[0] + " (rocprofv3 --pmc, gfx950-corrected)") if sm_traffic is not None
                                                 else "refused: sources changed since profiles/" + PMC_SUMMARY,
                         "workspace_bytes_per_launch": smk.get("workspace_bytes_per_launch"),
                         "algorithmic_bytes_per_launch": smk["algorithmic_bytes_per_launch"],
                         "share_of_step": sm_ms / s3["ms_per_batched_step"],
                         "device_copy_GBs_same_run": copy_gbs,
                         "note": "ISSUE / LATENCY-bound, not HBM-bound: 50 sweeps x 22 dependent block steps per launch, each ~75 "
                                 "instructions of ONE wave per component (a lone wave issues one VALU instruction per ~4.5 cycles) "
                                 "+ two LDS round trips, ~630 cycles; the mesh (47 KB) lives in LDS, the block inverses (176 KB "
                                 "per mesh, written once per launch) stream from L2 (workspace_bytes_per_launch: mostly L2 hits, "
                                 "see traffic). launch_ms = HIP events around the smoothing launches of the timed rollouts (the "
                                 "IPCS kernels of the previous env step run beside them); launch_ms_alone = the same meshes with "
                                 "nothing else on the chip. algorithmic bytes = coordinates in / out + cells; the HBM fraction is "
                                 "reported because the contract asks for it; it is not the resource that binds",
                         "traffic_measured_in_run": False,
                         "step_survey_convention_equivalent": {
                             "note": "NOT a roofline fraction: the bytes an assembled-CSR implementation would stream per batched "
                                     "step by SURVEY 8(d)'s convention, divided by the step time; the matrix-free kernels move "
                                     "~2 % of them (PMC)",
                             "survey_bytes_per_batched_step": s3_step_bytes, "s1_part_bytes_per_env": 0.7e6,
                             "ipcs_leg_bytes_survey_csr_convention": s3_ipcs_bytes, "survey_equivalent_GBs": s3_gbs,
                             "survey_equivalent_over_hbm_peak": s3_gbs / HBM_PEAK_GBS}},
            "roofline_s2_velocity": {
                "bound": "lds-atomic/fp64", "kernel": "at_velocity_kernel (rhs1 + matrix-free Jacobi-BiCGStab, LDS fp64 atomics)",
                "launch_ms": k_vel, "algorithmic_bytes_per_launch": vel_bytes, "achieved_GBs_model": vel_gbs,
                "frac_of_hbm_peak_model": vel_gbs / HBM_PEAK_GBS,
                "pmc_traffic_bytes_per_launch": vel_traffic,
                "frac_of_hbm_peak_pmc": None if not vel_traffic else vel_traffic / (k_vel * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernels_ms_per_step": {"at_velocity_kernel": k_vel, "at_pressure_kernel": k_prs, "at_correction_kernel": k_cor},
                "whole_step": {"implemented_bytes": step_bytes, "survey_csr_convention_bytes": survey_bytes},
                "fp64_valu": {"flops_per_launch": vel_flops, "achieved_TFLOPs": vel_flops / (k_vel * 1e-3) / 1e12,
                              "peak_TFLOPs_on_used_CUs": 78.6 * min(B, 256) / 256.0,
                              "frac": vel_flops / (k_vel * 1e-3) / 1e12 / (78.6 * min(B, 256) / 256.0)},
                "note": "matrix-free: operators are re-derived per triangle from 64 B of metadata, Krylov vectors live in "
                        "LDS/registers; one workgroup (one CU) per environment: 128 of 256 CUs at the BASELINE configuration"},
        }
        res["kernel_scratch"] = _kernel_scratch()
        res["rates"] = {"S3_north_star_step": s3, "S1_reference_step": s1, "S2_ipcs_step": s2, "S2_full_chip": full,
                        "training_loop": tr, **cfgs}
        if cpu is not None:
            res["cpu_baseline"] = cpu
        # the smoothing kernel against ITS bound (a dependency chain, not HBM): block steps x sweeps x 4 LDS round trips of
        # ~110 cycles each at the device clock
        clock_hz = MAX_CLOCK_HZ
        res["roofline"]["floor_ms"] = smk["block_steps_per_sweep"] * 50 * 4 * 110 / clock_hz * 1e3
        res["roofline"]["launch_over_floor"] = sm_ms / res["roofline"]["floor_ms"]
        detail_path = _write_detail(res)
        sys.stderr.write(_detail_digest(res) + "\n")
        line = json.dumps(compact_line(res, detail_path))
        assert len(line) <= LINE_LIMIT, len(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:               # the LAST thing this process writes to stdout
        print(line, file=real_stdout, flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
