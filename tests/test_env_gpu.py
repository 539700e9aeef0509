"""GPU parity: the reference-surface Env2DAirfoil (HIP interpolation / probes / smoothing) vs the oracle env."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

AGENT = dict(solver_steps=20, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1,
             u=-1, p=-1, time_reward=0.005, save_steps=4, goal_vertices=0.95, plot_dir="")


def _config(mesh):
    return dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"),
                                 geometry_params=dict(mesh=os.path.join(GOLDEN, f"{mesh}.npz")),
                                 solver_params=dict(dt=0.001, solver_type="lu", smooth=True, rtol=1e-12)),
                agent_params=dict(AGENT))


def _p2_to_oracle(u, n2):
    a = u.vector().get_local().reshape(n2, 2)
    return np.concatenate([a[:, 0], a[:, 1]])


@pytest.mark.parametrize("mesh", ["ys930", "ah93w145"])
def test_episode_matches_oracle(lib_built, meshes, mesh):
    from meshdqn_amd.env import Env2DAirfoil
    from oracle.env import OracleEnv
    env = Env2DAirfoil(_config(mesh))
    coords, cells = meshes[mesh]
    ora = OracleEnv(coords, cells, AGENT)
    # ground truth / snapshots of reset()
    assert np.allclose(env.gt_drag, ora.gt_drag, rtol=1e-8, atol=0)
    assert np.allclose(env.gt_lift, ora.gt_lift, rtol=1e-8, atol=0)
    n2 = ora.orig_th.np2
    for i in range(5):
        uo = ora.original_u[i]
        assert np.abs(_p2_to_oracle(env.original_u[i], n2) - uo).max() / np.abs(uo).max() < 1e-8
    s, so = env.get_state(), ora.get_state()
    assert s.x.shape == (180, 17) and s.x.dtype == torch.float32
    assert np.array_equal(s.edge_index.numpy(), so["edge_index"])
    assert np.allclose(s.x.numpy(), so["x"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(env.n_closest, ora.n_closest)       # bit-exact vertex ranking
    rng = np.random.default_rng(1370)
    for k in range(6):
        a = int(rng.integers(0, 181)) if k != 2 else 180       # include one "do nothing" action
        st, r, done, _ = env.step(a)
        sto, ro, doneo, _ = ora.step(a)
        assert done == doneo
        assert env.coord_map == ora.coord_map                   # removed-vertex indices bit-exact
        assert np.array_equal(env.n_closest, ora.n_closest)
        assert np.array_equal(env.flow_solver.mesh.cells(), ora.flow.mesh.cells)
        assert np.abs(env.flow_solver.mesh.coordinates() - ora.flow.mesh.coords).max() < 1e-12
        assert np.array_equal(st.edge_index.numpy(), sto["edge_index"])
        assert np.allclose(st.x.numpy(), sto["x"], rtol=1e-5, atol=1e-6)
        assert np.allclose(st.edge_attr, sto["edge_attr"], rtol=1e-12)
        if a != 180 or k > 0:
            assert np.allclose(env.new_drags, ora.new_drags, rtol=1e-8)
            assert np.allclose(env.new_lifts, ora.new_lifts, rtol=1e-8)
        assert abs(r - ro) < 1e-6 * max(1.0, abs(ro))
    # interpolated snapshots on the coarsened mesh
    n2c = ora.cur_th.np2
    for i in range(5):
        uo = ora.u[i]
        assert np.abs(_p2_to_oracle(env.u[i], n2c) - uo).max() / np.abs(uo).max() < 1e-8


def test_bad_actions_follow_reference_error_codes(lib_built):
    from meshdqn_amd.env import Env2DAirfoil
    env = Env2DAirfoil(_config("ah93w145"))
    env.get_state()
    st, r, done, _ = env.step(9999)  # KeyError in coord_map -> code 2: reward -1, terminal
    assert r == -1.0 and done is True


def test_deploy_resimulation_matches_oracle(lib_built, meshes, tmp_path):
    """deploy_dqn.py semantics: after every removal the operators are re-assembled on the coarsened mesh and the flow is
    re-simulated from rest; trajectories, complete_drags / complete_lifts and the final re-simulation (best mesh put
    back, re-meshed - smoothed - once more, :440-441,:495-517) in the reference's layouts.  Both orders of the work - the
    reference's (re-simulate inside the loop, one mesh at a time) and the batched one (policy first, all meshes as one
    IpcsBatch) - against the oracle, and against each other to 1e-9."""
    from meshdqn_amd.deploy import deploy
    from meshdqn_amd.env import Env2DAirfoil
    from oracle.env import OracleEnv
    from oracle.ipcs import OracleFlowSolver
    from oracle.mesh import OracleMesh
    coords, cells = meshes["ah93w145"]
    acts = [17, 180, 42, 3]
    outs = []
    for batched in (False, True):
        env = Env2DAirfoil(_config("ah93w145"))
        d = os.path.join(str(tmp_path), "batched" if batched else "sequential")
        # (the 20-step toy ground truth terminates episodes immediately: keep going to exercise the loop)
        outs.append(deploy(env, actions=acts, complete_traj=True, save_dir=d, prefix="t_", stop_on_done=False, batched=batched))
        assert len(env.flow_solver.mesh.coordinates()) == 794            # the last mesh is put back
        for k in ("interpolate_drag_trajectory", "drag_trajectory", "complete_drags", "complete_lifts", "actions"):
            assert os.path.exists(os.path.join(d, f"t_{k}.npy")), k
    seq, bat = outs
    assert seq["actions"].tolist() == acts and np.isnan(seq["selected"][1]) and bat["resimulated_meshes"] == 4
    for k in ("interpolate_drag_trajectory", "drag_trajectory", "complete_drags", "complete_lifts"):
        assert seq[k].shape == bat[k].shape and np.allclose(seq[k], bat[k], rtol=1e-9, atol=0), k
    assert abs(seq["new_drag"] - bat["new_drag"]) < 1e-9 * abs(seq["new_drag"])
    out = bat
    S = 5
    # rows: the initial mesh, then one per step (interpolated) / one per removal (re-simulated), [nv, S drags, S lifts]
    assert out["interpolate_drag_trajectory"].shape == (1 + len(acts), 1 + 2 * S)
    assert out["drag_trajectory"].shape == (1 + 3, 1 + 2 * S) and out["complete_drags"].shape == (1 + 3, S)
    assert out["traj_vertices"].tolist() == [797, 796, 795, 794] and out["est_vertices"].tolist() == [797, 796, 796, 795, 794]
    assert np.array_equal(out["drag_trajectory"][0, 1:1 + S], out["gt_drag"]) and np.array_equal(out["complete_lifts"][0], out["gt_lift"])
    # oracle: same removals, a fresh IPCS run from rest on every coarsened (already smoothed) mesh, and on the final mesh
    # smoothed once more
    ora = OracleEnv(coords, cells, AGENT)
    ora.get_state()
    row = 0
    for a in acts:
        ora.step(a)
        if a == 180:
            continue
        row += 1
        m = ora.flow.mesh
        fs = OracleFlowSolver(m.coords, m.cells, smooth=False)
        dr, li = [], []
        for i in range(AGENT["solver_steps"]):
            _, _, d, l = fs.evolve()
            if (i + 1) % AGENT["save_steps"] == 0:
                dr.append(d)
                li.append(l)
        assert np.allclose(out["traj_drag"][row], dr, rtol=1e-7) and np.allclose(out["complete_lifts"][row], li, rtol=1e-7)
    m = OracleMesh(ora.flow.mesh.coords.copy(), ora.flow.mesh.cells.copy()).smooth(50)
    fs = OracleFlowSolver(m.coords, m.cells, smooth=False)
    for i in range(AGENT["solver_steps"]):
        _, _, d, l = fs.evolve()
    assert abs(out["new_drag"] - d) < 1e-7 * abs(d)
    assert abs(out["drag_error_percent"] - 100 * abs(d - ora.gt_drag[-1]) / abs(ora.gt_drag[-1])) < 1e-4


def test_vec_env_matches_single_envs(lib_built):
    """Batched engine (C++ remesh + topology, batched GPU interpolation / probes) vs independent
    reference-surface environments (scipy Delaunay path) on per-environment action streams."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from meshdqn_amd.gcn_fused import FusedGcn
    from meshdqn_amd.data import Batch
    cfg = _config("ys930")
    base = Env2DAirfoil(cfg)
    B = 3
    venv = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2)
    singles = []
    for b in range(B):
        c = _config("ys930")
        c["agent_params"].update(gt_drag=base.gt_drag, gt_time=base.gt_time, u=base.original_u, p=base.original_p)
        e = Env2DAirfoil(c)
        e.gt_lift = base.gt_lift
        singles.append(e)
    st = venv.get_state()
    s_states = [e.get_state() for e in singles]
    rngs = [np.random.default_rng(1370 + b) for b in range(B)]
    net = NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    net = net.cuda()
    fused = FusedGcn(net)
    for k in range(5):
        for b in range(B):
            assert np.array_equal(st["n_closest"][b], singles[b].n_closest)
            assert st["coord_map"][b].tolist() == list(singles[b].coord_map.values())
            xs = s_states[b].x.numpy()
            assert np.allclose(st["x"][b].cpu().numpy(), xs, rtol=1e-5, atol=1e-6)
            e0, e1 = int(st["edge_ptr"][b]), int(st["edge_ptr"][b + 1])
            mine = sorted(zip(st["esrc"][e0:e1].tolist(), st["edst"][e0:e1].tolist()))
            ref = sorted(zip(*s_states[b].edge_index.tolist()))
            assert mine == ref  # same edges (cell ORDER is the engine's own)
        # Q-values of the batched arrays == Q-values of the per-env graphs (edge order does not matter)
        q = fused.forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], venv.N, venv.EMAX).cpu()
        qs = fused.forward(Batch.from_data_list([s.to("cuda") for s in s_states])).cpu()
        assert (q - qs).abs().max().item() < 1e-5
        acts = [int(r.integers(0, 181)) if (k + b) % 4 else 180 for b, r in enumerate(rngs)]
        st, rew, done, info = venv.step(acts)
        for b in range(B):
            s, r, d, _ = singles[b].step(acts[b])
            s_states[b] = s
            assert d == bool(done[b])
            assert abs(r - rew[b]) < 1e-6 * max(1.0, abs(r))
            if acts[b] != 180 or k > 0:
                assert np.allclose(info["new_drags"][b], singles[b].new_drags, rtol=1e-8)
            assert info["nv"][b] == len(singles[b].flow_solver.mesh.coordinates())


@pytest.mark.parametrize("pressure", ["cg", "direct"])
def test_vec_env_flow_step_matches_oracle(lib_built, pressure):
    """S3: after every batched remesh, IPCS steps on the coarsened meshes (host-engine index data + pattern-free GPU
    setup + matrix-free kernels, warm start = interpolated last snapshot) against the sparse-LU oracle solver built
    on the very same mesh and started from the same fields.  `direct`: the pressure matrix of every coarsened mesh is
    re-factorised on the device (mdq_ipcs_factorize_pressure; the reference re-factorises at every remesh,
    flow_solver.py:318-328): no Krylov iteration in the pressure solve."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    from oracle.ipcs import OracleFlowSolver
    cfg = _config("ys930")
    B, K = 2, 2
    venv = VecEnv2DAirfoil(cfg, B, base_env=Env2DAirfoil(cfg), auto_reset=False, nthreads=2, flow_steps=K, flow_rtol=1e-12,
                           flow_pressure=pressure)
    rngs = [np.random.default_rng(77 + b) for b in range(B)]
    venv.get_state()
    for k in range(3):
        acts = [int(r.integers(0, 180)) for r in rngs]
        st, rew, done, info = venv.step(acts)
        assert info["flow_drag"].shape == (B, K)
        it = venv.flow_iters.cpu().numpy()
        if pressure == "direct":
            assert (it[:, [0, 2]] > 0).all() and (it[:, 1] == 0).all() and (venv.flow_pd_status.cpu().numpy() == 0).all()
        else:
            assert (it > 0).all()
        for b in range(B):
            nv, nt = int(venv.nv[b]), int(venv.nt[b])
            ne = int(venv.h["ne"][b])
            n2 = nv + ne
            o = OracleFlowSolver(venv.coords[b, :nv].copy(), venv.cells[b, :nt].copy(), smooth=False)
            assert o.th.np2 == n2
            u0 = venv.u[b, venv.S - 1, :n2].cpu().numpy()
            o.u_n = np.concatenate([u0[:, 0], u0[:, 1]])
            o.p_n = venv.p[b, venv.S - 1, :nv].cpu().numpy().copy()
            for s in range(K):
                uo, po, do, lo = o.evolve()
                assert abs(info["flow_drag"][b, s] - do) < 1e-8 * abs(do), (k, b, s)
                assert abs(info["flow_lift"][b, s] - lo) < 1e-8 * abs(lo), (k, b, s)
            ug = venv.flow_t["u_n"][b, :n2].cpu().numpy()
            assert np.abs(np.concatenate([ug[:, 0], ug[:, 1]]) - uo).max() < 1e-8 * np.abs(uo).max()
            assert np.abs(venv.flow_t["p_n"][b, :nv].cpu().numpy() - po).max() < 1e-8 * np.abs(po).max()


def test_gpu_smoothing_matches_host_and_golden(lib_built, meshes):
    """mdq_smooth (level-scheduled Gauss-Seidel in LDS) vs the sequential host loop and the oracle's golden coordinates;
    untouched meshes (iterations 0) stay bit-identical; results are bitwise reproducible."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import smooth_batch_gpu
    from meshdqn_amd.topology import MeshTopology
    z = np.load(os.path.join(GOLDEN, "oracle_flow.npz"))
    names = ["ys930", "ah93w145", "ys930"]
    NV = max(meshes[n][0].shape[0] for n in names)
    NT = max(meshes[n][1].shape[0] for n in names)
    B = len(names)
    coords = np.zeros((B, NV, 2))
    cells = np.zeros((B, NT, 3), np.int32)
    nv = np.zeros(B, np.int32)
    nt = np.zeros(B, np.int32)
    for b, n in enumerate(names):
        c, t = meshes[n]
        coords[b, :len(c)], cells[b, :len(t)], nv[b], nt[b] = c, np.sort(t, axis=1), len(c), len(t)
    iters = np.array([50, 50, 0], np.int32)
    outs = []
    for rep in range(2):
        tc = torch.from_numpy(coords).cuda()
        smooth_batch_gpu(tc, torch.from_numpy(cells).cuda(), torch.from_numpy(nv).cuda(), torch.from_numpy(nt).cuda(),
                         torch.from_numpy(iters).cuda())
        torch.cuda.synchronize()
        outs.append(tc.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0][2], coords[2])
    for b, n in enumerate(names[:2]):
        host = smooth_coords(MeshTopology(*meshes[n]), 50)
        assert np.abs(outs[0][b, :nv[b]] - host).max() < 1e-13
        assert np.abs(outs[0][b, :nv[b]] - z[f"{n}_coords_smoothed"]).max() < 1e-13


def _fan_mesh(seed, n=12):
    """A hub of degree n (12) inside a strongly jittered ring of n interior vertices inside a boundary ring: exercises the
    exact fp64 path of mdq_smooth (degree > 8, step limited to half the minimum altitude)."""
    rng = np.random.default_rng(seed)
    ang = 2 * np.pi * np.arange(n) / n
    r1 = rng.uniform(0.25, 1.75, n)
    ring1 = np.stack([r1 * np.cos(ang + rng.uniform(-0.2, 0.2, n)), r1 * np.sin(ang)], 1)
    ring2 = 2.5 * np.stack([np.cos(ang + np.pi / n), np.sin(ang + np.pi / n)], 1)
    coords = np.concatenate([ring2, [[0.3, -0.2]], ring1])      # boundary ids first, like the reference meshes
    hub, i1, i2 = n, n + 1 + np.arange(n), np.arange(n)
    cells = []
    for j in range(n):
        k = (j + 1) % n
        cells += [[hub, i1[j], i1[k]], [i1[j], i2[j], i1[k]], [i1[j], i2[(j - 1) % n], i2[j]]]
    return coords, np.sort(np.array(cells, np.int32), axis=1)


def test_gpu_smoothing_exact_path(lib_built):
    """Limited steps and a vertex of degree 12: the fp32 fast decision must hand these updates to the exact fp64 path.
    Oracle = the plain DOLFIN loop (oracle/mesh.py), which is also asked how many steps were limited."""
    from meshdqn_amd.mesh_ops import smooth_batch_gpu
    from oracle.mesh import OracleMesh
    limited_total = 0
    for seed in range(6):
        coords, cells = _fan_mesh(seed)
        m = OracleMesh(coords, cells)
        assert (~m.on_boundary).sum() == 13 and max(len(n) for n in m.nbrs) == 12
        # count the limited steps of the first sweep with the oracle's own formulae
        x = m.coords.copy()
        for v in np.nonzero(~m.on_boundary)[0]:
            c = np.mean([x[w] for w in m.nbrs[v]], axis=0)
            cross = lambda e, w: e[0] * w[1] - e[1] * w[0]  # noqa: E731
            rmin = min(abs(cross(x[o[1]] - x[o[0]], x[v] - x[o[0]])) / np.linalg.norm(x[o[1]] - x[o[0]])
                       for o in ([m.cells[cc, j] for j in range(3) if j != k] for cc, k in m.vcells[v]))
            r = np.linalg.norm(c - x[v])
            limited_total += r > 0.5 * rmin
            x[v] = x[v] + min(r, 0.5 * rmin) * (c - x[v]) / r
        for iters in (1, 7):
            ref = OracleMesh(coords, cells).smooth(iters).coords
            tc = torch.from_numpy(coords[None].copy()).cuda()
            one = lambda v: torch.tensor([v], dtype=torch.int32, device="cuda")  # noqa: E731
            smooth_batch_gpu(tc, torch.from_numpy(cells[None].copy()).cuda(), one(len(coords)), one(len(cells)), one(iters))
            assert np.abs(tc[0].cpu().numpy() - ref).max() < 1e-12, (seed, iters)
    assert limited_total >= 6, "the synthetic meshes no longer exercise the limited step"


def _coarsened(meshes, name, removals, seed):
    """`name` smoothed, then `removals` interior vertices removed (host engine, smoothed after each); the LAST removal is
    left unsmoothed: the state the smoothing kernels meet inside an env step."""
    from meshdqn_amd.mesh_ops import remesh_batch
    from meshdqn_amd.topology import MeshTopology
    c0, t0 = meshes[name]
    coords = np.asarray(c0, np.float64)[None].copy()
    cells = np.sort(np.asarray(t0), axis=1).astype(np.int32)[None].copy()
    nv, nt = np.array([coords.shape[1]], np.int32), np.array([cells.shape[1]], np.int32)
    assert remesh_batch(coords, cells, nv, nt, np.array([-1], np.int32), 50)[0] == 0
    rng = np.random.default_rng(seed)
    for k in range(removals):
        interior = np.flatnonzero(~MeshTopology(coords[0, :nv[0]], cells[0, :nt[0]]).on_boundary)
        assert remesh_batch(coords, cells, nv, nt, np.array([int(rng.choice(interior))], np.int32),
                            50 if k < removals - 1 else 0)[0] == 0
    return coords[0, :nv[0]].copy(), cells[0, :nt[0]].copy()


def _smooth_both(batch, iters):
    """The same batch of meshes through mdq_smooth_fast and mdq_smooth; returns (fast, walk, diagnostics per env, nv)."""
    from meshdqn_amd.mesh_ops import smooth_batch_gpu, smooth_fast_stats
    B = len(batch)
    NV, NT = max(len(c) for c, _ in batch), max(len(t) for _, t in batch)
    coords, cells = np.zeros((B, NV, 2)), np.zeros((B, NT, 3), np.int32)
    nv, nt = np.zeros(B, np.int32), np.zeros(B, np.int32)
    for b, (c, t) in enumerate(batch):
        coords[b, :len(c)], cells[b, :len(t)], nv[b], nt[b] = c, t, len(c), len(t)
    dev = lambda a: torch.from_numpy(a).cuda()   # noqa: E731
    outs = []
    for fast in (True, False):
        tc = dev(coords.copy())
        smooth_batch_gpu(tc, dev(cells), dev(nv), dev(nt), dev(np.asarray(iters, np.int32)), fast=fast)
        torch.cuda.synchronize()
        outs.append(tc.cpu().numpy())
    return outs[0], outs[1], smooth_fast_stats(tc.device, B, NV), nv


def test_fast_smoothing_equals_the_walk_on_env_step_meshes(lib_built, meshes):
    """mdq_smooth_fast (blocked triangular solves, checked + repaired while limited steps occur, then validated in
    parallel) against mdq_smooth (the per-vertex walk, exact sequential semantics) on the meshes an env step produces: both
    airfoils after 1 .. 40 removals, the last removal not yet smoothed; 50, 4 and 3 iterations, and an untouched mesh.
    Equal to round-off (different association of the same sums), nothing handed back to the walk, a few repaired sweeps at
    most (the cavity's neighbours), bitwise reproducible."""
    batch = [_coarsened(meshes, n, k, 100 + k) for n in ("ys930", "ah93w145") for k in (1, 2, 9, 40)]
    iters = [50] * len(batch)
    iters[1], iters[2], iters[5] = 4, 3, 0
    fast, walk, redo, nv = _smooth_both(batch, iters)
    for b in range(len(batch)):
        assert np.abs(fast[b, :nv[b]] - walk[b, :nv[b]]).max() < 1e-13, b
    assert np.array_equal(fast[5], walk[5]) and (redo[:, 0] == 0).all() and (redo[:, 1] <= 3).all() and redo[:, 1].sum() > 0, redo
    again, _, _, _ = _smooth_both(batch, iters)
    assert np.array_equal(fast, again)
    # and against the sequential host loop (the oracle-pinned twin)
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.topology import MeshTopology
    host = smooth_coords(MeshTopology(*batch[0]), 50)
    assert np.abs(fast[0, :nv[0]] - host).max() < 1e-13


def test_fast_smoothing_rolls_back_sweeps_with_limited_steps(lib_built, meshes):
    """Interior vertices thrown far off their place: the step limit (half the smallest altitude) binds for many sweeps,
    so the validation of the blocked solve must reject those sweeps and redo them with the offending vertices on the
    exact update (repair rounds) - same result as the walk on its own, to round-off; the well-behaved mesh of the same
    launch needs no more than the repairs of its last removal."""
    from meshdqn_amd.topology import MeshTopology
    rng = np.random.default_rng(3)
    batch = []
    for n, moved in (("ys930", 6), ("ys930", 0), ("ah93w145", 25), ("ah93w145", 1)):   # (0.97: limited steps through ~7 sweeps)
        c, t = _coarsened(meshes, n, 3, 7)
        topo = MeshTopology(c, t)
        interior = np.flatnonzero(~topo.on_boundary)
        for v in rng.choice(interior, moved, replace=False):
            # move the vertex 97 % of the way towards one of the vertices it shares a cell with: thin, valid cells
            cellsv = t[(t == v).any(axis=1)]
            w = int([u for u in cellsv[0] if u != v][0])
            c[v] = c[v] + 0.97 * (c[w] - c[v])
        batch.append((c, t))
    fast, walk, redo, nv = _smooth_both(batch, [50] * len(batch))
    for b in range(len(batch)):
        assert np.abs(fast[b, :nv[b]] - walk[b, :nv[b]]).max() < 1e-12, b
    assert (redo[:, 0] == 0).all(), redo                              # nothing handed back: repaired in the kernel
    assert redo[0, 1] >= 4 and redo[2, 1] >= 4 and redo[3, 1] >= 3 and redo[1, 1] <= 2, redo   # sweeps with repair rounds


def test_fast_smoothing_hands_meshes_beyond_its_limits_to_the_walk(lib_built):
    """A hub of 18 cells (more than the 16 cells / 14 gather slots per row the blocked solve is laid out for): the kernel
    leaves the mesh untouched and reports all sweeps as handed back; the per-vertex walk (one workgroup over the handed-back
    environments) does them - same result as mdq_smooth on its own, bitwise; the ordinary mesh beside it in the same
    launch runs in the blocked solve."""
    from meshdqn_amd.mesh_ops import smooth_batch_gpu, smooth_fast_stats
    big = _fan_mesh(1, n=18)
    small = _fan_mesh(2)
    NV, NT = len(big[0]), len(big[1])
    coords, cells = np.zeros((2, NV, 2)), np.zeros((2, NT, 3), np.int32)
    nv, nt = np.array([len(big[0]), len(small[0])], np.int32), np.array([len(big[1]), len(small[1])], np.int32)
    for b, (c, t) in enumerate((big, small)):
        coords[b, :len(c)], cells[b, :len(t)] = c, t
    dev = lambda a: torch.from_numpy(a).cuda()   # noqa: E731
    outs = []
    for fast in (True, False):
        tc = dev(coords.copy())
        smooth_batch_gpu(tc, dev(cells), dev(nv), dev(nt), dev(np.array([5, 5], np.int32)), fast=fast)
        torch.cuda.synchronize()
        outs.append(tc.cpu().numpy())
    st = smooth_fast_stats(tc.device, 2, NV)
    assert st[0, 0] == 5 and st[1, 0] == 0, st
    assert np.array_equal(outs[0][0], outs[1][0])                       # handed back: the walk's own result
    assert np.abs(outs[0][1, :nv[1]] - outs[1][1, :nv[1]]).max() < 1e-13


def test_env_groups_equal_one_batch(lib_built):
    """G concurrently stepped groups (threads + streams) give the same trajectories as one batch stepped in place."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil, VecEnvGroups
    cfg = _config("ys930")
    base = Env2DAirfoil(cfg)
    B, G, K = 4, 2, 4
    one = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2, flow_steps=1)
    grp = VecEnvGroups(cfg, B, G, base_env=base, auto_reset=False, nthreads=2, flow_steps=1)
    script = np.random.default_rng(5).integers(0, 180, size=(K, B))
    one.get_state()
    ref = []
    for k in range(K):
        _, rew, done, info = one.step(script[k])
        ref.append((rew.copy(), info["new_drags"].copy(), info["flow_drag"].copy(), info["nv"].copy()))
    counters = [0] * G
    got = [[] for _ in range(G)]

    def act(g, env, st):
        k = counters[g]
        counters[g] += 1
        if k > 0:
            got[g].append((env.new_drags.copy(), env.flow_drag.copy(), env.nv.copy()))
        lo = sum(e.B for e in grp.envs[:g])
        return script[k, lo:lo + env.B]

    out = grp.rollout(act, K)
    for g, env in enumerate(grp.envs):
        got[g].append((env.new_drags.copy(), env.flow_drag.copy(), env.nv.copy()))
        lo = sum(e.B for e in grp.envs[:g])
        for k in range(K):
            nd, fd, nv = got[g][k]
            assert np.array_equal(nv, ref[k][3][lo:lo + env.B])
            assert np.allclose(nd, ref[k][1][lo:lo + env.B], rtol=1e-12, atol=0)
            assert np.allclose(fd, ref[k][2][lo:lo + env.B], rtol=1e-9, atol=0)
        assert np.allclose(out[g][0], ref[-1][0][lo:lo + env.B], rtol=1e-9)


@pytest.mark.parametrize("name", ["ys930", "ah93w145"])
def test_device_topology_engine_is_bit_identical_to_host_engine(lib_built, meshes, name):
    """mdq_env_topology (one workgroup per mesh in LDS) against mdq_env_topology_host, every output array, on the
    smoothed mesh, after vertex removals and with a shifted selection window; includes the IPCS index data."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import DeviceTopologyBatch, HostTopologyBatch, remesh_batch
    from meshdqn_amd.topology import MeshTopology
    coords, cells = meshes[name]
    t0 = MeshTopology(coords, cells)
    x0 = smooth_coords(t0, 50)
    tags = t0.facet_tags(x0)
    polygon = x0[[v for v in range(t0.nv) if t0.on_boundary[v] and -0.5 < x0[v, 0] < 3 and -0.5 < x0[v, 1] < 0.5]]
    B = 3
    args = (B, t0.nv, t0.nt, t0.ne, int((tags == 1).sum()), 180, 1536, polygon)
    hb = HostTopologyBatch(*args, ipcs=True)
    for b in range(B):
        hb.coords[b], hb.cells[b], hb.nv[b], hb.nt[b] = x0, np.sort(cells, axis=1), t0.nv, t0.nt
    interior = np.flatnonzero(~t0.on_boundary)
    for rnd in range(3):
        rem = np.array([-1, interior[40 + 7 * rnd], interior[300 + 11 * rnd]], np.int32)
        assert (remesh_batch(hb.coords, hb.cells, hb.nv, hb.nt, rem, 50, 2) == 0).all()
    hb.offset[:] = [0, 3, 0]
    hb.run(2)
    db = DeviceTopologyBatch(*args, device="cuda", ipcs=True, nse1_cap=hb.NSE1)
    db.coords.copy_(torch.from_numpy(hb.coords)); db.cells.copy_(torch.from_numpy(hb.cells))
    db.nv.copy_(torch.from_numpy(hb.nv)); db.nt.copy_(torch.from_numpy(hb.nt)); db.offset.copy_(torch.from_numpy(hb.offset))
    db.run()
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in db.t.items()}
    gi = {k: v.cpu().numpy() for k, v in db.ti.items()}
    for b in range(B):
        nv, nt, ne = int(hb.nv[b]), int(hb.nt[b]), int(hb.h["ne"][b])
        n2 = nv + ne
        for k in ("ne", "naf", "nremovable", "nsel", "nedges"):
            assert g[k][b] == hb.h[k][b], (k, b)
        naf, nE = int(hb.h["naf"][b]), int(hb.h["nedges"][b])
        assert np.array_equal(g["cell_dofs"][b][:, :nt], hb.h["cell_dofs"][b][:, :nt])
        assert np.array_equal(g["points"][b][:n2], hb.h["points"][b][:n2])                 # bitwise
        assert np.array_equal(g["af_facets"][b][:naf], hb.h["af_facets"][b][:naf])
        assert np.array_equal(g["n_closest"][b], hb.h["n_closest"][b])
        assert np.array_equal(g["coord_map"][b], hb.h["coord_map"][b])
        for k in ("edge_src", "edge_dst", "edge_len"):
            assert np.array_equal(g[k][b][:nE], hb.h[k][b][:nE]), (k, b)
        hi = hb.hi
        assert gi["nbo"][b] == hi["nbo"][b]
        nbo = int(hi["nbo"][b])
        nbe = int(hi["bo_ptr"][b][nbo])
        checks = dict(cell_outflow=nt, bcu_flag=n2, bcu_gx=n2, bcp_flag=nv, bo_rows=nbo, bo_ptr=nbo + 1, bo_col=nbe,
                      bo_src=nbe, g1_ptr=nv + 1, g1_src=3 * nt, g2_ptr=n2 + 1, g2_src=6 * nt, sl1_off=(nv + 63) // 64 + 1)
        for k, n in checks.items():
            assert np.array_equal(gi[k][b][:n], hi[k][b][:n]), (k, b)
        assert np.array_equal(gi["mf_scat"][b][:, :nt], hi["mf_scat"][b][:, :nt])
        nse = int(hi["sl1_off"][b][(nv + 63) // 64])
        assert np.array_equal(gi["sl1_col"][b][:nse], hi["sl1_col"][b][:nse])
    # the flow stream's variant: no selection / state graph, edge numbering taken from the first engine's cell dofs instead
    # of the hash table - the same index data, cell dofs, points and facets
    fb = DeviceTopologyBatch(*args, device="cuda", ipcs=True, nse1_cap=hb.NSE1, flow_only=True)
    fb.coords.copy_(db.coords); fb.cells.copy_(db.cells); fb.nv.copy_(db.nv); fb.nt.copy_(db.nt)
    fb.take_edges_from(db.t["cell_dofs"].clone(), db.t["ne"].clone())
    fb.run()
    torch.cuda.synchronize()
    for k in ("ne", "naf", "cell_dofs", "points", "af_facets"):
        assert torch.equal(fb.t[k], db.t[k]), k
    for k in db.ti:
        assert torch.equal(fb.ti[k], db.ti[k]), k
    bad = db.t["cell_dofs"].clone()
    bad[1, 3:] += 5000                                     # (not the dofs of this mesh: refused, not trusted)
    fb.take_edges_from(bad, db.t["ne"].clone())
    with pytest.raises(Exception):
        fb.run()


@pytest.mark.parametrize("name", ["ys930", "ah93w145"])
def test_device_remesh_matches_host_engine_over_an_episode(lib_built, meshes, name):
    """mdq_remesh + mdq_smooth against mdq_remesh_host (itself pinned to scipy Delaunay on the CPU suite): 30
    consecutive removals on 4 meshes with different action streams, incl. "do nothing" and a boundary vertex."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import remesh_batch, remesh_batch_gpu, smooth_batch_gpu
    from meshdqn_amd.topology import MeshTopology
    coords, cells = meshes[name]
    t0 = MeshTopology(coords, cells)
    x0 = smooth_coords(t0, 50)
    B, NV, NT = 4, t0.nv, t0.nt
    hc = np.repeat(x0[None], B, 0).copy()
    ht = np.repeat(np.sort(cells, axis=1)[None].astype(np.int32), B, 0).copy()
    hnv = np.full(B, NV, np.int32); hnt = np.full(B, NT, np.int32)
    dc, dtri = torch.from_numpy(hc).cuda(), torch.from_numpy(ht).cuda()
    dnv, dnt = torch.from_numpy(hnv).cuda(), torch.from_numpy(hnt).cuda()
    dst = torch.zeros(B, dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(11)
    for step in range(30):
        rem = np.empty(B, np.int32)
        for b in range(B):
            t = MeshTopology(hc[b, :hnv[b]], ht[b, :hnt[b]])
            interior = np.flatnonzero(~t.on_boundary)
            rem[b] = rng.choice(interior)
        if step % 7 == 3:
            rem[1] = -1                       # do nothing
        if step == 5:
            rem[2] = 0                        # a boundary vertex: must be refused, mesh untouched
        hst = remesh_batch(hc, ht, hnv, hnt, rem, 50, 2)
        drem = torch.from_numpy(rem).cuda()
        remesh_batch_gpu(dc, dtri, dnv, dnt, drem, dst)
        its = torch.where((drem >= 0) & (dst == 0), 50, 0).to(torch.int32)
        smooth_batch_gpu(dc, dtri, dnv, dnt, its)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy() != 0, hst != 0), (step, dst.cpu().numpy(), hst)
        assert np.array_equal(dnv.cpu().numpy(), hnv) and np.array_equal(dnt.cpu().numpy(), hnt)
        gc, gt = dc.cpu().numpy(), dtri.cpu().numpy()
        for b in range(B):
            mine = {tuple(r) for r in gt[b, :hnt[b]].tolist()}
            assert mine == {tuple(r) for r in ht[b, :hnt[b]].tolist()}, (step, b)
            assert np.abs(gc[b, :hnv[b]] - hc[b, :hnv[b]]).max() < 1e-12


def test_vec_env_host_and_device_engines_agree(lib_built):
    """The batched env with the C++ host engine, with GPU smoothing only, with the device topology engine and fully
    device-resident (default): same trajectories (identical selections, forces to 1e-9)."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = _config("ys930")
    base = Env2DAirfoil(cfg)
    B, K = 5, 6
    script = np.random.default_rng(21).integers(0, 181, size=(K, B))
    script[2, 1] = 180
    runs = []
    for kw in (dict(gpu_smoothing=False, gpu_topology=False, gpu_remesh=False), dict(gpu_topology=False, gpu_remesh=False),
               dict(gpu_remesh=False), dict()):
        env = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2, flow_steps=1, **kw)
        env.get_state()
        out = []
        for k in range(K):
            st, rew, done, info = env.step(script[k])
            out.append((info["nv"].copy(), st["coord_map"].copy(), st["nedges"].copy(), info["new_drags"].copy(),
                        info["flow_drag"].copy(), rew.copy(), done.copy(), st["x"].cpu().numpy()))
        runs.append(out)
    ref = runs[0]
    for run in runs[1:]:
        for a, r in zip(run, ref):
            assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1]) and np.array_equal(a[2], r[2])
            assert np.allclose(a[3], r[3], rtol=1e-9, atol=0) and np.allclose(a[4], r[4], rtol=1e-8, atol=0)
            assert np.allclose(a[5], r[5], rtol=1e-9) and np.array_equal(a[6], r[6])
            assert np.allclose(a[7], r[7], rtol=1e-5, atol=1e-6)


def test_device_resident_env_soak(lib_built):
    """80 batched steps of 32 environments under a random / greedy policy with auto-reset: finite outputs, vertex
    counts inside [0.95 nv0 - 1, nv0], state graphs well-formed, no engine failure codes."""
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.gcn_fused import FusedGcn
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = _config("ys930")
    venv = VecEnv2DAirfoil(cfg, 32, base_env=Env2DAirfoil(cfg), flow_steps=1)
    net = NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    fused = FusedGcn(net.cuda())
    rng = np.random.default_rng(0)
    st = venv.get_state()
    finished = 0
    for k in range(80):
        q = fused.forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], venv.N, venv.EMAX)
        assert torch.isfinite(q).all()
        acts = np.where(rng.random(32) < 0.7, rng.integers(0, 181, 32), q.argmax(1).cpu().numpy())
        st, rew, done, info = venv.step(acts)
        assert np.isfinite(rew).all() and np.isfinite(info["new_drags"]).all() and np.isfinite(info["flow_drag"]).all()
        assert (info["nv"] <= venv.NV).all() and (info["nv"] >= 0.95 * venv.NV - 2).all()
        assert (st["nedges"] % 3 == 0).all() and (st["nsel"] == 180).all() and (info["code"] == 0).all()
        finished += int(done.sum())
    assert finished > 0


def test_restore_rows_overwrites_exactly_the_listed_rows(lib_built):
    """mdq_restore_rows through the C ABI: mixed dtypes / row sizes, duplicate-free index list, bad arguments."""
    import ctypes as C
    from meshdqn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    tens = [torch.rand((9, 7, 3), dtype=torch.float64, generator=g).cuda(), torch.randint(0, 99, (9, 5), dtype=torch.int32, generator=g).cuda(),
            torch.rand((9, 1), dtype=torch.float32, generator=g).cuda()]
    srcs = [torch.rand((7, 3), dtype=torch.float64, generator=g).cuda(), torch.randint(0, 99, (5,), dtype=torch.int32, generator=g).cuda(),
            torch.rand((1,), dtype=torch.float32, generator=g).cuda()]
    before = [t.clone() for t in tens]
    idx = torch.tensor([6, 0, 3], dtype=torch.int32, device="cuda")
    n = len(tens)
    dst = (C.c_void_p * n)(*[t.data_ptr() for t in tens])
    src = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    nb = (C.c_int64 * n)(*[t[0].numel() * t.element_size() for t in tens])
    _lib.check(lib.mdq_restore_rows(n, dst, src, nb, 3, idx.data_ptr(), _lib.stream_ptr()), "mdq_restore_rows")
    torch.cuda.synchronize()
    for t, s, b in zip(tens, srcs, before):
        for r in range(9):
            assert torch.equal(t[r], s if r in (6, 0, 3) else b[r])
    assert lib.mdq_restore_rows(n, dst, src, nb, 0, None, _lib.stream_ptr()) == 0          # empty list: nothing to do
    assert lib.mdq_restore_rows(17, dst, src, nb, 3, idx.data_ptr(), _lib.stream_ptr()) != 0
    bad = (C.c_int64 * n)(6, 20, 4)                                                            # 6 bytes: not a multiple of 4
    assert lib.mdq_restore_rows(n, dst, src, bad, 3, idx.data_ptr(), _lib.stream_ptr()) != 0
    assert b"4-byte" in lib.mdq_last_error()


def test_compact_edges_matches_mask_indexing(lib_built):
    """mdq_compact_edges through the C ABI: packed edge lists == boolean-mask indexing of the padded arrays."""
    from meshdqn_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    B, EM = 7, 96
    cnt = rng.integers(0, EM + 1, size=B); cnt[3] = 0; cnt[5] = EM
    sp = torch.from_numpy(rng.integers(0, 180, size=(B, EM)).astype(np.int32)).cuda()
    dp = torch.from_numpy(rng.integers(0, 180, size=(B, EM)).astype(np.int32)).cuda()
    ep = np.zeros(B + 1, np.int32); ep[1:] = np.cumsum(cnt)
    ep_d = torch.from_numpy(ep).cuda()
    es = torch.full((int(ep[-1]),), -7, dtype=torch.int32, device="cuda"); ed = es.clone()
    _lib.check(lib.mdq_compact_edges(B, EM, sp.data_ptr(), dp.data_ptr(), ep_d.data_ptr(), es.data_ptr(), ed.data_ptr(),
                                     _lib.stream_ptr()), "mdq_compact_edges")
    torch.cuda.synchronize()
    live = torch.from_numpy(np.arange(EM)[None, :] < cnt[:, None]).cuda()
    assert torch.equal(es, sp[live]) and torch.equal(ed, dp[live])
    assert lib.mdq_compact_edges(0, EM, sp.data_ptr(), dp.data_ptr(), ep_d.data_ptr(), es.data_ptr(), ed.data_ptr(), _lib.stream_ptr()) != 0


@pytest.mark.parametrize("name", ["ys930", "ah93w145"])
def test_gpu_smoothing_is_bitwise_reproducible(lib_built, meshes, name):
    """The smoothing kernel hands positions from pass to pass through LDS (in-order service of one wave's operations,
    slots of a level filled in arrival order): any stale read or order dependence would show up as a difference
    between environments or launches."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import smooth_batch_gpu
    from meshdqn_amd.topology import MeshTopology
    coords, cells = meshes[name]
    ref = smooth_coords(MeshTopology(coords, cells), 50)
    B = 192
    c0 = torch.from_numpy(np.repeat(np.asarray(coords, np.float64)[None], B, 0).copy()).cuda()
    ct = torch.from_numpy(np.repeat(np.sort(np.asarray(cells), 1)[None].astype(np.int32), B, 0).copy()).cuda()
    nv = torch.full((B,), len(coords), dtype=torch.int32, device="cuda")
    nt = torch.full((B,), len(cells), dtype=torch.int32, device="cuda")
    it = torch.full((B,), 50, dtype=torch.int32, device="cuda")
    first = None
    for _ in range(6):
        c = c0.clone()
        smooth_batch_gpu(c, ct, nv, nt, it)
        out = c.cpu().numpy()
        assert (out == out[0]).all()
        first = out[0].copy() if first is None else first
        assert (out[0] == first).all()
    assert np.abs(first - ref).max() < 1e-13


def test_reference_driver_call_sequences(lib_built, tmp_path, monkeypatch):
    """The reference drivers' own call sequences against meshdqn_amd (the drop-in boundary of SURVEY 8b):
    airfoil_dqn.py:394-401 (main process: env, set_plot_dir, plot_state, ground truth into the config), :435/:448 (the
    workers rebuild the env from that config: the snapshot-reload branch Env2DAirfoil.py:126-153), and
    deploy_dqn.py:84-92 (plot_state with a title, flow_solver.deploy(), snapshots handed over as functions)."""
    from meshdqn_amd.env import Env2DAirfoil
    monkeypatch.chdir(tmp_path)
    flow_config = _config("ys930")
    flow_config["agent_params"].update(solver_steps=200, save_steps=40)
    save_dir = "training_results/run0"
    # ---- airfoil_dqn.py:394-401
    env = Env2DAirfoil(flow_config)
    env.set_plot_dir(save_dir)
    written = env.plot_state()
    assert os.path.isfile(written) and os.path.dirname(written).endswith(save_dir)
    flow_config["agent_params"]["plot_dir"] = save_dir
    flow_config["agent_params"]["gt_drag"] = env.gt_drag
    flow_config["agent_params"]["gt_time"] = env.gt_time
    n_actions = flow_config["agent_params"]["N_closest"]
    assert n_actions == env.action_space.n == 180
    for f in ("velocities.npy", "pressures.npy", "save_velocities.npy", "save_pressures.npy"):
        assert os.path.isfile(os.path.join(save_dir, "snapshots", f))
    # ---- airfoil_dqn.py:435,448: a worker's env (u = p = -1 in the yaml: snapshots come back from the files)
    worker = Env2DAirfoil(flow_config)
    assert len(worker.original_u) == len(env.original_u) == 5
    for a, b in zip(worker.original_u + worker.original_p, env.original_u + env.original_p):
        assert torch.equal(a.data, b.data)
    assert np.array_equal(worker.gt_drag, env.gt_drag)
    s_w, s_e = worker.get_state(), env.get_state()
    assert torch.equal(s_w.x, s_e.x) and torch.equal(s_w.edge_index, s_e.edge_index)
    out_w, out_e = worker.step(7), env.step(7)
    assert out_w[1] == out_e[1] and out_w[2] == out_e[2] and torch.equal(out_w[0].x, out_e[0].x)
    assert np.array_equal(worker.new_drags, env.new_drags)
    # ---- deploy_dqn.py:84-92
    dcfg = _config("ys930")
    dcfg["agent_params"].update(solver_steps=200, save_steps=40, plot_dir=save_dir)
    denv = Env2DAirfoil(dcfg)
    assert os.path.isfile(denv.plot_state(title="{} Closest Vertices to Airfoil", filename="deploy_state"))
    denv.flow_solver.deploy()
    dcfg["agent_params"]["gt_drag"] = denv.gt_drag
    dcfg["agent_params"]["gt_time"] = denv.gt_time
    dcfg["agent_params"]["u"] = [u.copy(deepcopy=True) for u in denv.original_u]
    dcfg["agent_params"]["p"] = [p.copy(deepcopy=True) for p in denv.original_p]
    assert denv.action_space.n == 180
    env2 = Env2DAirfoil(dcfg)                        # deploy_dqn.py:299-300 style re-creation from the functions
    assert torch.equal(env2.original_u[-1].data, denv.original_u[-1].data)
    gt = env2.return_vals()
    assert np.array_equal(gt[0], denv.gt_drag)


def test_interpolation_failure_restores_the_solver(lib_built, monkeypatch):
    """Env2DAirfoil.py:556-558: when the interpolation onto the new mesh fails the solver goes back to the old mesh
    (and here to everything derived from it), the step reports code 2 = terminal with the negative reward."""
    from meshdqn_amd.env import Env2DAirfoil
    cfg = _config("ys930")
    cfg["agent_params"].update(solver_steps=100, save_steps=20)
    env = Env2DAirfoil(cfg)
    env.get_state()
    old_mesh, old_removable = env.flow_solver.mesh, list(env.flow_solver.removable)
    nv0 = len(old_mesh.coordinates())

    def boom(*a, **k):
        raise RuntimeError("no cell found")
    monkeypatch.setattr(env._interp, "interpolate", boom)
    st, rew, done, _ = env.step(5)
    assert rew == -1.0 and done
    assert env.flow_solver.mesh is old_mesh and len(env.flow_solver.mesh.coordinates()) == nv0
    assert list(env.flow_solver.removable) == old_removable
    assert st.x.shape == (180, 17) and env.velocities.shape[1] == nv0


def test_overlapped_flow_step_equals_inline_flow_step(lib_built):
    """S3 with the IPCS step on a second stream beside the next env step (double-buffered topology outputs, forces
    delivered one step later) vs the in-line IPCS step: same states, rewards, dones, and the SAME flow forces shifted
    by one step (both are the same kernels on the same data; only the stream and the reporting step differ)."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = _config("ys930")
    cfg["agent_params"].update(solver_steps=200, save_steps=40)
    base = Env2DAirfoil(cfg)
    B, K = 6, 9
    envs = [VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=1, flow_rtol=1e-12, flow_overlap=ov) for ov in (False, True)]
    rng = np.random.default_rng(3)
    acts = [[int(rng.integers(0, 181)) for _ in range(B)] for _ in range(K)]
    for e in envs:
        e.get_state()
    hist = [[], []]
    for k in range(K):
        for i, e in enumerate(envs):
            st, rew, done, info = e.step(acts[k])
            hist[i].append((st["x"].cpu().numpy(), rew.copy(), done.copy(), info))
    for k in range(K):
        a, b = hist[0][k], hist[1][k]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert a[3]["flow_lag"] == 0 and b[3]["flow_lag"] == 1
        if k == 0:
            assert b[3]["flow_drag"] is None
        else:
            assert np.allclose(b[3]["flow_drag"], hist[0][k - 1][3]["flow_drag"], rtol=1e-6, atol=0)   # (LDS fp64 atomics: not bitwise)
            assert np.allclose(b[3]["flow_lift"], hist[0][k - 1][3]["flow_lift"], rtol=1e-6, atol=0)
    last = envs[1].flow_wait()
    assert np.allclose(last[0], hist[0][K - 1][3]["flow_drag"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("flow", [0, 1])
def test_device_resident_rollout_equals_host_logic_steps(lib_built, flow):
    """`rollout_device` (action decoding, smoothing request, reward / terminal logic and in-place resets as kernels:
    mdq_env_act / mdq_env_smooth_iters / mdq_env_result / mdq_restore_rows_masked, one read-back at the end) against the
    same number of `step()` calls (host logic of Env2DAirfoil.py:318-428) - scripted actions incl. "do nothing" (180),
    invalid actions and enough removals in some environments to reach the terminal condition and the in-place reset;
    then with the fused Q-network choosing (epsilon-greedy from given random streams)."""
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.gcn_fused import FusedGcn
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = _config("ys930")
    cfg["agent_params"].update(solver_steps=200, save_steps=40, goal_vertices=0.992)      # 8 removals end an episode
    base = Env2DAirfoil(cfg)
    B, K = 8, 14
    host = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=flow, flow_overlap=bool(flow))
    devi = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=flow, flow_overlap=bool(flow))
    rng = np.random.default_rng(12)
    acts = rng.integers(0, 181, size=(K, B))
    acts[3, 0], acts[5, 1], acts[:, 2] = 180, 180, 180            # window shifts; env 2 never removes anything
    host.get_state()
    ref = [host.step(acts[k]) for k in range(K)]
    out = devi.rollout_device(None, K, actions=acts)
    assert out["dones"].any() and (out["codes"] == 0).all()
    for k in range(K):
        _, rew, done, info = ref[k]
        assert np.array_equal(out["dones"][k], done), k
        assert np.array_equal(out["nv"][k], info["nv"]), k
        assert np.array_equal(out["codes"][k], info["code"]), k
        assert np.abs(out["rewards"][k] - rew).max() < 1e-12, k
    # the host mirrors follow the device: the two environments are in the same state and continue identically
    for a in ("nv", "nt", "offset", "steps"):
        assert np.array_equal(getattr(devi, a), getattr(host, a)), a
    nvm = int(host.nv.max())
    assert np.array_equal(devi.coords[:, :nvm], host.coords[:, :nvm]) or np.abs(devi.coords[:, :nvm] - host.coords[:, :nvm]).max() < 1e-15
    sh, sd = host.get_state(), devi.get_state()
    assert torch.equal(sh["x"], sd["x"]) and torch.equal(sh["esrc"], sd["esrc"]) and torch.equal(sh["edge_ptr"], sd["edge_ptr"])
    # policy-driven: the fused Q-network + given random streams on both paths
    torch.manual_seed(1)
    net = NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    fused = FusedGcn(net.cuda())
    explore = rng.random((6, B)) < 0.5
    rand = rng.integers(0, 181, size=(6, B))
    st = host.get_state()
    hist = []
    for k in range(6):
        q = fused.forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], host.N, host.EMAX)
        a = np.where(explore[k], rand[k], q.argmax(1).cpu().numpy())
        st, rew, done, info = host.step(a)
        hist.append((a, rew, done))
    out = devi.rollout_device(fused, 6, explore=explore, rand_actions=rand)
    for k in range(6):
        assert np.array_equal(out["actions"][k], hist[k][0]), k
        assert np.abs(out["rewards"][k] - hist[k][1]).max() < 1e-12 and np.array_equal(out["dones"][k], hist[k][2])


def test_stream_calibration_keeps_a_working_flow_stream_and_resets_the_envs(lib_built):
    """`VecEnv2DAirfoil.calibrate_streams` (a few real steps per candidate flow stream, the fastest stays) leaves the
    environments in their initial state and the overlapped flow path working: the rollout that follows equals the one
    of an uncalibrated twin action by action; `streams.concurrent_stream` hands out a stream that is not the current one."""
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.gcn_fused import FusedGcn
    from meshdqn_amd.streams import concurrent_stream
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = _config("ys930")
    base = Env2DAirfoil(cfg)
    B, K = 6, 5
    torch.manual_seed(0)
    net = NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    net.cuda()
    outs = []
    for calibrate in (False, True):
        venv = VecEnv2DAirfoil(cfg, B, base_env=base, nthreads=2, flow_steps=1, flow_overlap=True)
        fg = FusedGcn(net)
        if calibrate:
            ms = venv.calibrate_streams(fg, tries=3, steps=3)
            # (an empty list: this process has already chosen a flow stream for this main stream - streams.LOG says how)
            assert len(ms) <= 3 and all(m > 0 for m in ms)
            assert venv.calibrate_streams(fg, tries=3, steps=3) == []      # once per process and main stream
            assert (venv.nv == venv.NV).all() and (venv.steps == 0).all() and (venv.offset == 0).all()
        rng = np.random.default_rng(3)
        outs.append(venv.rollout_device(fg, K, rng.random((K, B)) < 0.5, rng.integers(0, 181, (K, B))))
        venv.flow_wait()
    a, b = outs
    assert np.array_equal(a["actions"], b["actions"]) and np.array_equal(a["dones"], b["dones"]) and np.array_equal(a["nv"], b["nv"])
    assert np.allclose(a["rewards"], b["rewards"], rtol=1e-9, atol=1e-12)
    s = concurrent_stream("cuda")
    assert isinstance(s, torch.cuda.Stream) and s != torch.cuda.current_stream()
    from meshdqn_amd import streams
    r = streams.role_streams("cuda")
    assert r is streams.role_streams("cuda:0") and len({id(v) for v in r.values()}) == 3 and venv._flow_stream is not None
    assert any(e["event"].startswith("roles created") for e in streams.LOG)
    assert any(e["event"] == "flow stream calibrated" for e in streams.LOG)


@pytest.mark.parametrize("kind", ["star", "far", "sliver", "dense", "ring"])
def test_device_closest_ranking_equals_host_engine_for_other_polygons(lib_built, meshes, kind):
    """The N-closest ranking of `mdq_env_topology` (fp32 estimates pick the polygon segments that can hold a vertex's
    minimum distance and the segments its ray straddles; only those are evaluated exactly) against the host engine's full
    fp64 loop, bit for bit, for polygons that stress the selection: a star that swallows part of the mesh (vertices inside:
    distance 0 and ties broken by index), a small polygon far outside (every distance large, all segments nearly
    equidistant), a sliver with near-degenerate segments, a dense polygon with the capacity's 256 points, a regular polygon centred on a
    vertex (all segments equidistant: the candidate queue overflows)."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import DeviceTopologyBatch, HostTopologyBatch
    from meshdqn_amd.topology import MeshTopology
    coords, cells = meshes["ys930"]
    t0 = MeshTopology(coords, cells)
    x0 = smooth_coords(t0, 50)
    tags = t0.facet_tags(x0)
    rng = np.random.default_rng(5)
    if kind == "star":
        th = np.sort(rng.uniform(0, 2 * np.pi, 97))
        r = 0.12 + 0.1 * rng.random(97)
        polygon = np.stack([0.6 + r * np.cos(th), 0.2 + 0.7 * r * np.sin(th)], axis=1)
    elif kind == "far":
        th = np.linspace(0, 2 * np.pi, 41, endpoint=False)
        polygon = np.stack([40.0 + 1e-3 * np.cos(th), -25.0 + 1e-3 * np.sin(th)], axis=1)
    elif kind == "sliver":
        xs = np.linspace(0.3, 1.9, 60)
        polygon = np.concatenate([np.stack([xs, 0.2 + 1e-9 * np.sin(40 * xs)], axis=1),
                                  np.stack([xs[::-1], 0.2 + 1e-7 + 1e-9 * np.cos(33 * xs[::-1])], axis=1),
                                  [[0.3, 0.2 + 5e-8], [0.3, 0.2 + 5e-8]]])        # (a repeated point: a degenerate segment)
    elif kind == "ring":
        # a regular 64-gon centred ON an interior vertex: for that vertex every segment is equally far (more candidates than
        # the kernel's four-slot queue holds: its every-segment path), its neighbours sit near the centre too
        c = x0[np.flatnonzero(~t0.on_boundary)[400]]
        th = np.linspace(0, 2 * np.pi, 64, endpoint=False)
        polygon = np.stack([c[0] + 0.11 * np.cos(th), c[1] + 0.11 * np.sin(th)], axis=1)
    else:
        th = np.linspace(0, 2 * np.pi, 256, endpoint=False)
        polygon = np.stack([1.1 + 0.5 * np.cos(th) * (1 + 0.05 * np.sin(9 * th)), 0.2 + 0.1 * np.sin(th)], axis=1)
    B = 2
    args = (B, t0.nv, t0.nt, t0.ne, int((tags == 1).sum()), 180, 1536, np.ascontiguousarray(polygon))
    hb = HostTopologyBatch(*args)
    for b in range(B):
        hb.coords[b], hb.cells[b], hb.nv[b], hb.nt[b] = x0, np.sort(cells, axis=1), t0.nv, t0.nt
    hb.coords[1] += 1e-7 * rng.standard_normal(hb.coords[1].shape) * (~t0.on_boundary)[:, None]     # (a second, jittered mesh)
    hb.offset[:] = [0, 5]
    hb.run(2)
    db = DeviceTopologyBatch(*args, device="cuda")
    db.coords.copy_(torch.from_numpy(hb.coords)); db.cells.copy_(torch.from_numpy(hb.cells))
    db.nv.copy_(torch.from_numpy(hb.nv)); db.nt.copy_(torch.from_numpy(hb.nt)); db.offset.copy_(torch.from_numpy(hb.offset))
    db.run()
    g = {k: v.cpu().numpy() for k, v in db.t.items()}
    for b in range(B):
        for k in ("nremovable", "nsel", "nedges"):
            assert g[k][b] == hb.h[k][b], (kind, k, b)
        assert np.array_equal(g["n_closest"][b], hb.h["n_closest"][b]), (kind, b)
        assert np.array_equal(g["coord_map"][b], hb.h["coord_map"][b]), (kind, b)
        nE = int(hb.h["nedges"][b])
        for k in ("edge_src", "edge_dst", "edge_len"):
            assert np.array_equal(g[k][b][:nE], hb.h[k][b][:nE]), (kind, k, b)
