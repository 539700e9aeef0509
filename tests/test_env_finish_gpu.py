"""GPU: the fused launches of the device-resident env step (round 4) against the entry points they replace, through the
C ABI on random data: `mdq_env_finish` == mdq_copy_strided (hand-over) -> mdq_env_result -> mdq_restore_rows_masked ->
mdq_state_features, and `mdq_remesh_act` == mdq_env_act -> mdq_remesh.  (The env-level tests - stock episodes, rollout
vs step - run through the fused launches as well.)"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _finish_case(seed, B, N, S, NV, NP, auto_reset, with_handover, with_xinit):
    from meshdqn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(seed)
    dev = "cuda"
    f64, i32 = torch.float64, torch.int32
    F = 2 + 3 * S
    rnd = lambda *sh: torch.rand(sh, dtype=f64, generator=g)                     # noqa: E731
    gt = -(0.1 + rnd(S))
    drags = gt[None] * (1.0 + 2e-3 * (rnd(B, S) - 0.5))                          # errors around the 1e-3 threshold
    nv = torch.randint(int(0.93 * NV), NV + 1, (B,), generator=g, dtype=i32)
    rstat = (torch.rand(B, generator=g) < 0.1).to(i32) * -3
    tstat = (torch.rand(B, generator=g) < 0.05).to(i32) * 7
    nsel = torch.where(torch.rand(B, generator=g) < 0.1, torch.tensor(N - 3, dtype=i32), torch.tensor(N, dtype=i32))
    code = (torch.rand(B, generator=g) < 0.1).to(i32) * 2
    steps = torch.randint(0, 12, (B,), generator=g, dtype=i32)
    u, p, coords = rnd(B, S, NP, 2), rnd(B, S, NV), rnd(B, NV, 2)
    ncl = torch.randint(0, NV, (B, N), generator=g, dtype=i32)
    cells = torch.randint(0, NV, (B, 2 * NV, 3), generator=g, dtype=i32)
    misc = torch.randint(0, 99, (B, 5), generator=g, dtype=i32)                   # a row that is not 16-byte sized
    extra = torch.randint(0, 99, (B, 6, 7), generator=g, dtype=i32)               # handed over, never reset
    src = dict(u=rnd(S, NP, 2), p=rnd(S, NV), coords=rnd(NV, 2), cells=torch.randint(0, NV, (2 * NV, 3), generator=g, dtype=i32),
               misc=torch.randint(0, 99, (5,), generator=g, dtype=i32), nv=torch.tensor([NV], dtype=i32),
               ncl=torch.randint(0, NV, (N,), generator=g, dtype=i32), nsel=torch.tensor([N], dtype=i32))
    rows = dict(u=u, p=p, coords=coords, cells=cells, misc=misc, nv=nv.clone(), ncl=ncl, nsel=nsel.clone())
    xinit = torch.rand((N, F), generator=g)

    def to_dev(d_):
        return {k: v.to(dev).contiguous() for k, v in d_.items()}
    sp = _lib.stream_ptr
    out = []
    for fused in (False, True):
        r, s_ = to_dev(rows), to_dev(src)
        ex = extra.to(dev)
        t = dict(gt=gt.to(dev), drags=drags.to(dev), rstat=rstat.to(dev), tstat=tstat.to(dev), code_in=code.to(dev),
                 steps_in=steps.to(dev), xinit=xinit.to(dev))
        rew = torch.zeros(B, dtype=f64, device=dev)
        done = torch.zeros(B, dtype=torch.uint8, device=dev)
        err = torch.zeros(1, dtype=i32, device=dev)
        nvo = torch.zeros(B, dtype=i32, device=dev)
        x = torch.zeros((B, N, F), dtype=torch.float32, device=dev)
        ho = dict(coords=torch.zeros_like(r["coords"]), u_n=torch.zeros((B, NP, 2), dtype=f64, device=dev),
                  p_n=torch.zeros((B, NV), dtype=f64, device=dev), nv=torch.zeros(B, dtype=i32, device=dev), extra=torch.zeros_like(ex))
        keys = list(r)
        nv_in, nsel_in = r["nv"], r["nsel"]
        if not fused:
            if with_handover:
                for dst, s in ((ho["coords"], r["coords"]), (ho["u_n"], r["u"][:, S - 1]), (ho["p_n"], r["p"][:, S - 1]),
                               (ho["nv"], r["nv"]), (ho["extra"], ex)):
                    dst.copy_(s)
            code_io, steps_io = t["code_in"].clone(), t["steps_in"].clone()
            _lib.check(lib.mdq_env_result(B, N, S, t["drags"].data_ptr(), t["gt"].data_ptr(), nv_in.data_ptr(), NV, t["rstat"].data_ptr(),
                                          t["tstat"].data_ptr(), nsel_in.data_ptr(), code_io.data_ptr(), steps_io.data_ptr(), 1e-3,
                                          0.005, 0.95, 10, -1.0, int(auto_reset), rew.data_ptr(), done.data_ptr(), err.data_ptr(),
                                          nvo.data_ptr(), sp()), "mdq_env_result")
            if auto_reset:
                n = len(keys)
                dst = (C.c_void_p * n)(*[r[k].data_ptr() for k in keys])
                sr = (C.c_void_p * n)(*[s_[k].data_ptr() for k in keys])
                nb = (C.c_int64 * n)(*[r[k][0].numel() * r[k].element_size() for k in keys])
                _lib.check(lib.mdq_restore_rows_masked(n, dst, sr, nb, B, done.data_ptr(), sp()), "mdq_restore_rows_masked")
            _lib.check(lib.mdq_state_features(B, N, S, NV, NP, r["coords"].data_ptr(), r["u"].data_ptr(), r["p"].data_ptr(),
                                              r["ncl"].data_ptr(), r["nsel"].data_ptr(), x.data_ptr(), sp()), "mdq_state_features")
            if auto_reset and with_xinit:          # the fused launch shows the cached features for reset environments
                x[done.bool()] = t["xinit"]
        else:
            d = _lib.EnvFinishDesc()
            d.B, d.N, d.S, d.NV, d.NP = B, N, S, NV, NP
            d.nv0, d.timesteps, d.auto_reset = NV, 10, int(auto_reset)
            d.threshold, d.time_reward, d.goal_vertices, d.negative_reward = 1e-3, 0.005, 0.95, -1.0
            code_io, steps_io = torch.zeros(B, dtype=i32, device=dev), torch.zeros(B, dtype=i32, device=dev)
            d.new_drags, d.gt_drag, d.nv, d.rstat = t["drags"].data_ptr(), t["gt"].data_ptr(), nv_in.data_ptr(), t["rstat"].data_ptr()
            d.topo_status, d.nsel, d.code_in, d.code_out = t["tstat"].data_ptr(), nsel_in.data_ptr(), t["code_in"].data_ptr(), code_io.data_ptr()
            d.steps_in, d.steps_out = t["steps_in"].data_ptr(), steps_io.data_ptr()
            d.reward, d.done, d.err_flag, d.nv_out = rew.data_ptr(), done.data_ptr(), err.data_ptr(), nvo.data_ptr()
            n = len(keys)
            for i, k in enumerate(keys):
                d.dst[i], d.src[i] = r[k].data_ptr(), (s_[k].data_ptr() if auto_reset else None)
                d.row_bytes[i] = r[k][0].numel() * r[k].element_size()
            if with_handover:
                per_u, per_p = NP * 2 * 8, NV * 8
                for k, dst, off, nb in (("coords", ho["coords"], 0, NV * 16), ("u", ho["u_n"], (S - 1) * per_u, per_u),
                                        ("p", ho["p_n"], (S - 1) * per_p, per_p), ("nv", ho["nv"], 0, 4)):
                    i = keys.index(k)
                    d.handover_dst[i], d.handover_off[i], d.handover_bytes[i] = dst.data_ptr(), off, nb
                d.dst[n], d.src[n], d.row_bytes[n] = ex.data_ptr(), None, ex[0].numel() * 4
                d.handover_dst[n], d.handover_off[n], d.handover_bytes[n] = ho["extra"].data_ptr(), 0, ex[0].numel() * 4
                n += 1
            d.n_rows = n
            d.coords, d.u, d.p, d.n_closest = r["coords"].data_ptr(), r["u"].data_ptr(), r["p"].data_ptr(), r["ncl"].data_ptr()
            d.x_init = t["xinit"].data_ptr() if with_xinit else None
            d.x = x.data_ptr()
            arrive = torch.zeros(B, dtype=i32, device=dev)
            d.arrive = arrive.data_ptr()
            _lib.check(lib.mdq_env_finish(C.byref(d), sp()), "mdq_env_finish")
            assert int(arrive.abs().sum()) == 0           # the arrival counters are left at zero for the next launch
        torch.cuda.synchronize()
        out.append(dict(rew=rew.cpu(), done=done.cpu(), err=err.cpu(), nvo=nvo.cpu(), code=code_io.cpu(), steps=steps_io.cpu(), x=x.cpu(),
                        **{f"row_{k}": v.cpu() for k, v in r.items()}, **{f"ho_{k}": v.cpu() for k, v in ho.items()}))
    return out


@pytest.mark.parametrize("auto_reset,with_handover,with_xinit", [(True, True, True), (True, False, True), (False, True, False),
                                                                   (True, True, False), (False, False, False)])
def test_env_finish_equals_the_four_launches_it_replaces(lib_built, auto_reset, with_handover, with_xinit):
    for seed, (B, N, S, NV, NP) in enumerate([(37, 24, 5, 96, 300), (128, 180, 5, 876, 3476), (3, 7, 2, 33, 101)]):
        a, b = _finish_case(seed, B, N, S, NV, NP, auto_reset, with_handover, with_xinit)
        assert a["done"].sum() > 0 or B < 8
        for k in a:
            assert torch.equal(a[k], b[k]), (seed, k)          # bit for bit: rewards, flags, every restored / handed-over row


def test_env_finish_far_beyond_the_resident_workgroups(lib_built):
    """B Y = 24 576 workgroups against ~2 000 resident slots: the workgroups (b, y > 0) of an environment start long after (b, 0)
    has finished.  The terminal decision of every workgroup reads nv / nsel - rows the same launch resets in place - so a
    late workgroup must still see the PRE-reset values (round 4's kernel took a reset environment for a running one there and
    left its rows half restored; nv in [0.93, 1] NV: a third of the environments end on the goal_vertices criterion)."""
    a, b = _finish_case(11, 1536, 180, 5, 876, 3476, True, True, True)
    assert a["done"].sum() > 300
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_env_finish_refuses_bad_descriptors(lib_built):
    from meshdqn_amd import _lib
    lib = _lib.load()
    d = _lib.EnvFinishDesc()
    assert lib.mdq_env_finish(C.byref(d), _lib.stream_ptr()) != 0 and b"bad arguments" in lib.mdq_last_error()


@pytest.mark.parametrize("given", [False, True])
def test_remesh_act_equals_env_act_then_remesh(lib_built, meshes, given):
    """The action decoding as the head of the removal kernel: same actions / codes / window offsets / removed vertices, same
    meshes, for Q-rows with ties and NaNs, explored actions, "do nothing", out-of-range and out-of-vertices actions."""
    from meshdqn_amd import _lib
    lib = _lib.load()
    coords0, cells0 = meshes["ys930"]
    NV, NT, N, B = coords0.shape[0], cells0.shape[0], 180, 48
    rng = np.random.default_rng(5)
    i32 = torch.int32
    interior = np.flatnonzero(~__import__("meshdqn_amd.topology", fromlist=["MeshTopology"]).MeshTopology(coords0, cells0).on_boundary)
    cmap = torch.from_numpy(rng.choice(interior, (B, N)).astype(np.int32)).cuda()
    nsel = torch.from_numpy(np.where(rng.random(B) < 0.2, 170, N).astype(np.int32)).cuda()
    q = rng.standard_normal((B, N + 1)).astype(np.float32)
    q[3, 5] = q[3, 9] = q[3].max() + 1.0                 # a tie: the first maximum
    q[4] = np.nan
    q[5, N] = 99.0                                        # greedy "do nothing"
    q[6, 175] = 99.0
    explore = (rng.random(B) < 0.4).astype(np.uint8)
    rand = rng.integers(0, N + 1, B).astype(np.int32)
    rand[7], explore[7] = N, 1
    acts = rng.integers(0, N, B).astype(np.int32)
    acts[:6] = [-1, N, N + 1, 179, -5, N + 40]            # invalid, "do nothing", out of range, maybe beyond nsel, ...
    res = []
    for fused in (False, True):
        coords = torch.from_numpy(np.repeat(coords0[None], B, 0).copy()).cuda()
        cells = torch.from_numpy(np.repeat(np.sort(cells0, axis=1).astype(np.int32)[None], B, 0).copy()).cuda()
        nv, nt = torch.full((B,), NV, dtype=i32, device="cuda"), torch.full((B,), NT, dtype=i32, device="cuda")
        off = torch.arange(B, dtype=i32, device="cuda")
        action = torch.from_numpy(acts.copy()).cuda()
        rem, code, stat = (torch.full((B,), -7, dtype=i32, device="cuda") for _ in range(3))
        qd, ed, rd = torch.from_numpy(q).cuda(), torch.from_numpy(explore).cuda(), torch.from_numpy(rand).cuda()
        qa = (None, None, None) if given else (qd.data_ptr(), ed.data_ptr(), rd.data_ptr())
        sp = _lib.stream_ptr
        if fused:
            _lib.check(lib.mdq_remesh_act(B, NV, NT, coords.data_ptr(), cells.data_ptr(), nv.data_ptr(), nt.data_ptr(), N, *qa,
                                          nsel.data_ptr(), cmap.data_ptr(), off.data_ptr(), action.data_ptr(), rem.data_ptr(),
                                          code.data_ptr(), stat.data_ptr(), None, 0, sp()), "mdq_remesh_act")
        else:
            _lib.check(lib.mdq_env_act(B, N, *qa, nsel.data_ptr(), cmap.data_ptr(), off.data_ptr(), action.data_ptr(),
                                       rem.data_ptr(), code.data_ptr(), sp()), "mdq_env_act")
            _lib.check(lib.mdq_remesh(B, NV, NT, coords.data_ptr(), cells.data_ptr(), nv.data_ptr(), nt.data_ptr(), rem.data_ptr(),
                                      stat.data_ptr(), None, 0, sp()), "mdq_remesh")
        torch.cuda.synchronize()
        res.append([t.cpu() for t in (coords, cells, nv, nt, off, action, rem, code, stat)])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert (res[1][6] >= 0).sum() > B // 2 and (res[1][2] == NV - 1).sum() > B // 2 and (res[1][7] == 2).any() and (res[1][6] < 0).any()


def _venv(meshes_dir, B, flow_steps):
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"),
                                geometry_params=dict(mesh=os.path.join(meshes_dir, "ys930.npz")),
                                solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
               agent_params=dict(solver_steps=40, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1,
                                 u=-1, p=-1, time_reward=0.005, save_steps=8, goal_vertices=0.95, plot_dir=""))
    return VecEnv2DAirfoil(cfg, B, flow_steps=flow_steps, flow_overlap=bool(flow_steps))


@pytest.mark.parametrize("sparse", [1, 2])
def test_sparse_interpolation_equals_the_full_fields_where_a_rollout_reads_them(lib_built, sparse):
    """`mdq_interp_desc.sparse` (round 4: the device-resident step interpolates only what it reads): vertices at every
    snapshot, the edge midpoints of the airfoil-facet cells at every snapshot, every edge midpoint at the last snapshot
    (sparse = 1; sparse = 2: no other edge values at all) - bit for bit the values of the full interpolation, on meshes
    that have taken a few removals."""
    venv = _venv(GOLDEN, 6, 0)
    rng = np.random.default_rng(3)
    for _ in range(4):
        venv.step(rng.integers(0, 180, venv.B))
    dt, S = venv.dtopo, venv.S
    venv._refresh_launch(readback=False, sparse=0)
    u0, p0 = venv.u.clone(), venv.p.clone()
    venv._interp_bufs[venv._interp_i ^ 1][0].fill_(float("nan"))          # the set the next call writes
    venv._interp_bufs[venv._interp_i ^ 1][1].fill_(float("nan"))
    venv._refresh_launch(readback=False, sparse=sparse)
    torch.cuda.synchronize()
    u1, p1 = venv.u, venv.p
    nv, ne, naf = dt.nv.cpu().numpy(), dt.t["ne"].cpu().numpy(), dt.t["naf"].cpu().numpy()
    af, cd = dt.t["af_facets"].cpu().numpy(), dt.t["cell_dofs"].cpu().numpy()
    for b in range(venv.B):
        n1, n2 = int(nv[b]), int(nv[b] + ne[b])
        assert torch.equal(p1[b, :, :n1], p0[b, :, :n1])
        assert torch.equal(u1[b, :, :n1], u0[b, :, :n1])
        edges = np.unique(cd[b, 3:, af[b, :naf[b], 0]].reshape(-1))
        assert edges.size >= naf[b] and edges.min() >= n1 and edges.max() < n2
        assert torch.equal(u1[b, :, edges], u0[b, :, edges])
        if sparse == 1:
            assert torch.equal(u1[b, S - 1, :n2], u0[b, S - 1, :n2])
        other = np.setdiff1d(np.arange(n1, n2), edges)
        untouched = torch.isnan(u1[b, :S - 1][:, other]).all() if sparse == 1 else torch.isnan(u1[b][:, other]).all()
        assert bool(untouched)                                              # (nothing else was computed)


def test_topology_handover_outputs_equal_the_mesh_and_its_edge_numbering(lib_built):
    """`mdq_topo_handover` (round 4): the second output set of mdq_env_topology - what the flow stream's engine reads - holds
    the rows of the mesh and the cell dofs / edge counts the launch wrote to its own outputs."""
    venv = _venv(GOLDEN, 5, 0)
    rng = np.random.default_rng(4)
    for _ in range(3):
        venv.step(rng.integers(0, 180, venv.B))
    dt = venv.dtopo
    ho = dict(coords=torch.full_like(dt.coords, -1.0), cells=torch.full_like(dt.cells, -1), nv=torch.full_like(dt.nv, -1),
              nt=torch.full_like(dt.nt, -1), cell_dofs=torch.full_like(dt.t["cell_dofs"], -1), ne=torch.full_like(dt.t["ne"], -1))
    dt.set_handover(**ho)
    dt.run()
    dt.set_handover()
    torch.cuda.synchronize()
    assert torch.equal(ho["nv"], dt.nv) and torch.equal(ho["nt"], dt.nt) and torch.equal(ho["ne"], dt.t["ne"])
    for b in range(venv.B):
        n, t = int(dt.nv[b]), int(dt.nt[b])
        assert torch.equal(ho["coords"][b, :n], dt.coords[b, :n]) and torch.equal(ho["cells"][b, :t], dt.cells[b, :t])
        assert torch.equal(ho["cell_dofs"][b, :, :t], dt.t["cell_dofs"][b, :, :t])
    with pytest.raises(ValueError):
        dt.set_handover(**{**ho, "cells": ho["cells"][:, :-1]})


def test_fields_after_a_rollout_whose_last_step_resets_environments(lib_built):
    """`venv.u / venv.p` are public: after `rollout_device` the rows of the environments the LAST step reset in place hold the
    initial snapshot fields (what the reference's reset() exposes: the cached ground-truth snapshots), bit for bit - not an
    interpolation of pre-reset fields onto the restored mesh (the full interpolation pass of `rollout_end` runs with the restored
    vertex counts) -, and the rows of every other environment are the COMPLETE fields of their current meshes (equal to a
    full interpolation launched afterwards); a host-driven step() that follows continues from there."""
    venv = _venv(GOLDEN, 12, 0)
    venv.get_state()
    rng = np.random.default_rng(7)
    acts = rng.integers(0, 180, (3, venv.B)).astype(np.int32)
    out = venv.rollout_device(None, 3, actions=acts)
    torch.cuda.synchronize()
    last = out["dones"][-1]
    assert last.any() and not last.all(), last          # (threshold 1e-3: some episodes end at their third removal, not all)
    c = venv._init_cache
    u, p = venv.u.clone(), venv.p.clone()
    for b in np.flatnonzero(last):
        assert torch.equal(u[b], c["u"]) and torch.equal(p[b], c["p"]), b      # (the rows cached from environment 0 at reset_all)
        assert int(venv.nv[b]) == venv.NV and int(venv.steps[b]) == 0
    venv._refresh_launch(readback=False, sparse=0)       # a full interpolation of the current meshes
    torch.cuda.synchronize()
    nv, ne = venv.dtopo.nv.cpu().numpy(), venv.dtopo.t["ne"].cpu().numpy()
    for b in np.flatnonzero(~last):
        n1, n2 = int(nv[b]), int(nv[b] + ne[b])
        assert torch.equal(venv.u[b, :, :n2], u[b, :, :n2]) and torch.equal(venv.p[b, :, :n1], p[b, :, :n1]), b
    st, rew, done, info = venv.step(rng.integers(0, 180, venv.B))
    assert np.isfinite(rew).all() and (info["code"] == 0).all()
