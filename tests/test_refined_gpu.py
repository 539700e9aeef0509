"""GPU: the env step on a mesh beyond 1024 vertices (BASELINE configs[4]: ys930 red-refined once, 3 322 vertices / 6 280
triangles) - the large-mesh instances of the device kernels (round 4: mdq_remesh / mdq_env_topology with their tables on a
slab in global memory, the level-scheduled smoothing kernel) against the C++ host twins, which the CPU suite pins to
scipy / the oracle; then `VecEnv2DAirfoil` (device-resident) against the host-engine `VecEnv2DAirfoil` on the same scripts.
Reference: Env2DAirfoil._remove_vertex / _check_mesh (Env2DAirfoil.py:452-512, 547-602), FlowSolver.remesh
(flow_solver.py:233-359) take whatever mesh they are given."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _refined(meshes, name="ys930"):
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import red_refine
    from meshdqn_amd.topology import MeshTopology
    coords, cells = meshes[name]
    rc, rcells = red_refine(smooth_coords(MeshTopology(coords, cells), 50), cells)
    t = MeshTopology(rc, rcells)
    return t, smooth_coords(t, 50)


def test_large_mesh_remesh_and_smoothing_match_the_host_engine(lib_built, meshes):
    """12 consecutive removals on 3 refined meshes with different action streams (incl. "do nothing" and a boundary vertex):
    cells as a set, vertex counts and failure codes equal, smoothed coordinates to 1e-12."""
    from meshdqn_amd.mesh_ops import remesh_batch, remesh_batch_gpu, smooth_batch_gpu
    from meshdqn_amd.topology import MeshTopology
    t0, x0 = _refined(meshes)
    assert t0.nv == 3322 and t0.nt == 6280
    B, NV, NT = 3, t0.nv, t0.nt
    hc = np.repeat(x0[None], B, 0).copy()
    ht = np.repeat(np.sort(t0.cells, axis=1)[None].astype(np.int32), B, 0).copy()
    hnv = np.full(B, NV, np.int32); hnt = np.full(B, NT, np.int32)
    dc, dtri = torch.from_numpy(hc).cuda(), torch.from_numpy(ht).cuda()
    dnv, dnt = torch.from_numpy(hnv).cuda(), torch.from_numpy(hnt).cuda()
    dst = torch.zeros(B, dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(11)
    for step in range(12):
        rem = np.empty(B, np.int32)
        for b in range(B):
            t = MeshTopology(hc[b, :hnv[b]], ht[b, :hnt[b]])
            rem[b] = rng.choice(np.flatnonzero(~t.on_boundary))
        if step % 5 == 3:
            rem[1] = -1
        if step == 4:
            rem[2] = 0                        # a boundary vertex: refused, mesh untouched
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
            assert {tuple(r) for r in gt[b, :hnt[b]].tolist()} == {tuple(r) for r in ht[b, :hnt[b]].tolist()}, (step, b)
            assert np.abs(gc[b, :hnv[b]] - hc[b, :hnv[b]]).max() < 1e-12, (step, b)


def test_large_mesh_smoothing_is_reproducible_and_matches_the_host_loop(lib_built, meshes):
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import red_refine, smooth_batch_gpu
    from meshdqn_amd.topology import MeshTopology
    coords, cells = meshes["ah93w145"]
    rc, rcells = red_refine(coords, cells)                 # NOT smoothed before: limited steps in the first sweeps
    t = MeshTopology(rc, rcells)
    host = smooth_coords(t, 50)
    outs = []
    for rep in range(2):
        tc = torch.from_numpy(np.stack([rc, rc])).cuda()
        tt = torch.from_numpy(np.sort(rcells, axis=1).astype(np.int32)[None].repeat(2, 0).copy()).cuda()
        one = lambda v: torch.full((2,), v, dtype=torch.int32, device="cuda")     # noqa: E731
        its = torch.tensor([50, 7], dtype=torch.int32, device="cuda")
        smooth_batch_gpu(tc, tt, one(t.nv), one(t.nt), its)
        outs.append(tc.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0][0] - host).max() < 1e-13
    assert np.abs(outs[0][1] - smooth_coords(t, 7)).max() < 1e-13


def test_large_mesh_topology_engine_is_bit_identical_to_host_engine(lib_built, meshes):
    """mdq_env_topology's large-mesh instance against mdq_env_topology_host: every output array of the S1 step (edges, dof
    map, points, airfoil facets, removable vertices, N-closest window, state graph), after removals, with a shifted window."""
    from meshdqn_amd.mesh_ops import DeviceTopologyBatch, HostTopologyBatch, remesh_batch
    t0, x0 = _refined(meshes)
    tags = t0.facet_tags(x0)
    polygon = x0[[v for v in range(t0.nv) if t0.on_boundary[v] and -0.5 < x0[v, 0] < 3 and -0.5 < x0[v, 1] < 0.5]]
    assert 200 < len(polygon) <= 256
    B = 3
    args = (B, t0.nv, t0.nt, t0.ne, int((tags == 1).sum()), 180, 1536, polygon)
    hb = HostTopologyBatch(*args, ipcs=False)
    for b in range(B):
        hb.coords[b], hb.cells[b], hb.nv[b], hb.nt[b] = x0, np.sort(t0.cells, axis=1), t0.nv, t0.nt
    interior = np.flatnonzero(~t0.on_boundary)
    for rnd in range(3):
        rem = np.array([-1, interior[40 + 7 * rnd], interior[1300 + 11 * rnd]], np.int32)
        assert (remesh_batch(hb.coords, hb.cells, hb.nv, hb.nt, rem, 50, 2) == 0).all()
    hb.offset[:] = [0, 3, 0]
    hb.run(2)
    db = DeviceTopologyBatch(*args, device="cuda", ipcs=False)
    db.coords.copy_(torch.from_numpy(hb.coords)); db.cells.copy_(torch.from_numpy(hb.cells))
    db.nv.copy_(torch.from_numpy(hb.nv)); db.nt.copy_(torch.from_numpy(hb.nt)); db.offset.copy_(torch.from_numpy(hb.offset))
    db.run()
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in db.t.items()}
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


@pytest.mark.slow
def test_vec_env_steps_the_refined_mesh_on_the_device(lib_built, meshes, tmp_path):
    """`VecEnv2DAirfoil` on the red-refined ys930 (3 322 vertices): the device-resident engine (`step()` and
    `rollout_device`) against the same class on the C++ host engine, per-environment scripted actions incl. "do nothing":
    identical selections / vertex counts / terminal flags, rewards and forces to 1e-9."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    t0, _ = _refined(meshes)
    path = os.path.join(str(tmp_path), "ys930_refined.npz")
    np.savez(path, coords=t0.coords, cells=t0.cells)
    cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=path),
                                solver_params=dict(dt=0.001, solver_type="lu", smooth=True, rtol=1e-10)),
               agent_params=dict(solver_steps=10, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1,
                                 gt_time=-1, u=-1, p=-1, time_reward=0.005, save_steps=2, goal_vertices=0.95, plot_dir=""))
    base = Env2DAirfoil(cfg)
    assert len(base.flow_solver.mesh.coordinates()) == 3322
    B, K = 3, 5
    script = np.random.default_rng(21).integers(0, 181, size=(K, B))
    script[2, 1] = 180
    runs = []
    for kw in (dict(gpu_smoothing=False, gpu_topology=False, gpu_remesh=False), dict()):
        env = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2, **kw)
        env.get_state()
        out = []
        for k in range(K):
            st, rew, done, info = env.step(script[k])
            out.append((info["nv"].copy(), st["coord_map"].copy(), st["nedges"].copy(), info["new_drags"].copy(), rew.copy(),
                        done.copy(), st["x"].cpu().numpy()))
        runs.append(out)
    for a, r in zip(runs[1], runs[0]):
        assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1]) and np.array_equal(a[2], r[2])
        assert np.allclose(a[3], r[3], rtol=1e-9, atol=0) and np.allclose(a[4], r[4], rtol=1e-9) and np.array_equal(a[5], r[5])
        assert np.allclose(a[6], r[6], rtol=1e-5, atol=1e-6)
    assert (runs[1][-1][0] < 3322).all()
    # and without a host round trip inside the steps
    dev = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2)
    dev.get_state()
    out = dev.rollout_device(None, K, actions=script)
    for k in range(K):
        assert np.array_equal(out["nv"][k], runs[0][k][0]) and np.array_equal(out["dones"][k], runs[0][k][5])
        assert np.abs(out["rewards"][k] - runs[0][k][4]).max() < 1e-9


def test_large_mesh_ipcs_index_data_is_bit_identical_to_host_engine(lib_built, meshes):
    """The index data of the matrix-free IPCS path from the large-mesh topology instance (Dirichlet flags / values, outflow
    cells and facet entries, dof <- element-slot lists, SELL pattern of the P1 Laplacian) against the host engine - what the
    S3 step on the refined mesh feeds to mdq_ipcs_setup_matfree and to mode 5 of mdq_ipcs_evolve."""
    from meshdqn_amd.mesh_ops import DeviceTopologyBatch, HostTopologyBatch, remesh_batch
    t0, x0 = _refined(meshes)
    tags = t0.facet_tags(x0)
    polygon = x0[[v for v in range(t0.nv) if t0.on_boundary[v] and -0.5 < x0[v, 0] < 3 and -0.5 < x0[v, 1] < 0.5]]
    B = 2
    nbo_cap = 2 * (2 * int((tags == 3).sum()) + 1)
    args = (B, t0.nv, t0.nt, t0.ne, int((tags == 1).sum()), 180, 1536, polygon)
    hb = HostTopologyBatch(*args, ipcs=True, nbo_cap=nbo_cap)
    for b in range(B):
        hb.coords[b], hb.cells[b], hb.nv[b], hb.nt[b] = x0, np.sort(t0.cells, axis=1), t0.nv, t0.nt
    interior = np.flatnonzero(~t0.on_boundary)
    rem = np.array([-1, interior[700]], np.int32)
    assert (remesh_batch(hb.coords, hb.cells, hb.nv, hb.nt, rem, 50, 2) == 0).all()
    hb.run(2)
    db = DeviceTopologyBatch(*args, device="cuda", ipcs=True, nse1_cap=hb.NSE1, nbo_cap=nbo_cap)
    db.coords.copy_(torch.from_numpy(hb.coords)); db.cells.copy_(torch.from_numpy(hb.cells))
    db.nv.copy_(torch.from_numpy(hb.nv)); db.nt.copy_(torch.from_numpy(hb.nt)); db.offset.copy_(torch.from_numpy(hb.offset))
    db.run()
    torch.cuda.synchronize()
    gi = {k: v.cpu().numpy() for k, v in db.ti.items()}
    hi = hb.hi
    for b in range(B):
        nv, nt, ne = int(hb.nv[b]), int(hb.nt[b]), int(hb.h["ne"][b])
        n2 = nv + ne
        assert gi["nbo"][b] == hi["nbo"][b] and hi["nbo"][b] > 20
        nbo = int(hi["nbo"][b])
        nbe = int(hi["bo_ptr"][b][nbo])
        checks = dict(cell_outflow=nt, bcu_flag=n2, bcu_gx=n2, bcp_flag=nv, bo_rows=nbo, bo_ptr=nbo + 1, bo_col=nbe,
                      bo_src=nbe, g1_ptr=nv + 1, g1_src=3 * nt, g2_ptr=n2 + 1, g2_src=6 * nt, sl1_off=(nv + 63) // 64 + 1)
        for k, n in checks.items():
            assert np.array_equal(gi[k][b][:n], hi[k][b][:n]), (k, b)
        nse = int(hi["sl1_off"][b][(nv + 63) // 64])
        assert np.array_equal(gi["sl1_col"][b][:nse], hi["sl1_col"][b][:nse])


@pytest.mark.slow
def test_vec_env_s3_step_on_the_refined_mesh(lib_built, meshes, tmp_path):
    """The north-star step on the refined mesh: after every batched removal IPCS steps on every coarsened mesh (device
    topology with index data -> mdq_ipcs_setup_matfree -> mode 5 of mdq_ipcs_evolve through the dof <- slot lists, warm start
    = interpolated last snapshot) against the sparse-LU oracle solver on the very same mesh and start fields; and the same
    step with the IPCS leg on the flow stream (results one step late) against the in-line one."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    from oracle.ipcs import OracleFlowSolver
    t0, _ = _refined(meshes)
    path = os.path.join(str(tmp_path), "ys930_refined.npz")
    np.savez(path, coords=t0.coords, cells=t0.cells)
    cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=path),
                                solver_params=dict(dt=0.001, solver_type="lu", smooth=True, rtol=1e-10)),
               agent_params=dict(solver_steps=10, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1,
                                 gt_time=-1, u=-1, p=-1, time_reward=0.005, save_steps=2, goal_vertices=0.95, plot_dir=""))
    base = Env2DAirfoil(cfg)
    B, K = 2, 2
    acts = np.array([[5, 40], [180, 77]])
    venv = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2, flow_steps=K, flow_rtol=1e-12)
    venv.get_state()
    inline = []
    for k in range(2):
        st, rew, done, info = venv.step(acts[k])
        inline.append((info["flow_drag"].copy(), info["flow_lift"].copy()))
        assert (venv.flow_iters.cpu().numpy() > 0).all()
        for b in range(B):
            nv, nt = int(venv.nv[b]), int(venv.nt[b])
            n2 = nv + int(venv.h["ne"][b])
            o = OracleFlowSolver(venv.coords[b, :nv].copy(), venv.cells[b, :nt].copy(), smooth=False)
            assert o.th.np2 == n2
            u0 = venv.u[b, venv.S - 1, :n2].cpu().numpy()
            o.u_n = np.concatenate([u0[:, 0], u0[:, 1]])
            o.p_n = venv.p[b, venv.S - 1, :nv].cpu().numpy().copy()
            for s_ in range(K):
                uo, po, do, lo = o.evolve()
                assert abs(info["flow_drag"][b, s_] - do) < 1e-8 * abs(do), (k, b, s_)
                assert abs(info["flow_lift"][b, s_] - lo) < 1e-8 * abs(lo), (k, b, s_)
    over = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2, flow_steps=K, flow_rtol=1e-12, flow_overlap=True)
    over.get_state()
    over.step(acts[0])
    _, _, _, info = over.step(acts[1])
    assert info["flow_lag"] == 1 and np.allclose(info["flow_drag"], inline[0][0], rtol=1e-9)     # the flow of step 0, delivered late
    last = over.flow_wait()
    assert np.allclose(last[0], inline[1][0], rtol=1e-9) and np.allclose(last[1], inline[1][1], rtol=1e-9)


def test_large_mesh_entry_points_use_the_callers_workspace_only(lib_built, meshes):
    """include/meshdqn_hip.h: "the library performs no hidden allocation ... returns without a device sync" - round 4's
    large-mesh instances grew process-static slabs with hipMalloc (+ hipStreamSynchronize / hipFree) inside mdq_remesh,
    mdq_env_topology and mdq_smooth*.  ABI 6: the tables live in a caller-owned workspace.  Here, through the C ABI: (1) every
    entry point REFUSES a missing / short / misaligned workspace with a negative code and a text; (2) after a one-mesh warm-up
    on the same stream (code objects loaded, the queue's private-segment backing allocated by the runtime - a NEW stream would
    make the runtime allocate that for its queue while a kernel with scratch runs, which is not the library's doing) a call
    with EIGHT meshes leaves hipMemGetInfo's free byte count where it was - the old slabs were grown on demand, i.e. exactly this
    call allocated (and synchronised the stream to free the smaller slab); (3) the queries return 0 for the 1024-vertex kernels and -1 beyond the 4096-vertex ones."""
    import ctypes as C
    from meshdqn_amd import _lib
    from meshdqn_amd.mesh_ops import DeviceTopologyBatch
    lib = _lib.load()
    t0, x0 = _refined(meshes)
    NV, NT = t0.nv, t0.nt
    assert lib.mdq_remesh_workspace_bytes(4, 876, 1570) == 0 and lib.mdq_smooth_workspace_bytes(4, 876, 1570) == 0
    # (mdq_remesh: 16 384 vertices since round 6 - the hash on the slab too; topology and smoothing: 4 096)
    assert lib.mdq_remesh_workspace_bytes(4, 5000, 9000) > 4 * 2_000_000 and lib.mdq_remesh_workspace_bytes(4, 17000, 9000) == -1
    assert lib.mdq_remesh_workspace_bytes(4, 5000, 33000) == -1 and lib.mdq_smooth_workspace_bytes(4, 5000, 9000) > 4 * 2_000_000
    assert lib.mdq_smooth_workspace_bytes(4, 17000, 9000) == -1 and lib.mdq_smooth_fast_workspace_bytes(4, 17000) == -1
    tags = t0.facet_tags(x0)
    polygon = x0[[v for v in range(t0.nv) if t0.on_boundary[v] and -0.5 < x0[v, 0] < 3 and -0.5 < x0[v, 1] < 0.5]]
    interior = np.flatnonzero(~t0.on_boundary)
    i32 = torch.int32

    def state(B):
        c = torch.from_numpy(np.repeat(x0[None], B, 0).copy()).cuda()
        t = torch.from_numpy(np.repeat(np.sort(t0.cells, axis=1)[None].astype(np.int32), B, 0).copy()).cuda()
        return dict(c=c, t=t, nv=torch.full((B,), NV, dtype=i32, device="cuda"), nt=torch.full((B,), NT, dtype=i32, device="cuda"),
                    rem=torch.from_numpy(interior[100:100 + B].astype(np.int32)).cuda(), st=torch.zeros(B, dtype=i32, device="cuda"),
                    its=torch.full((B,), 3, dtype=i32, device="cuda"))

    def calls(B, s, ws):
        """mdq_remesh, mdq_smooth, mdq_smooth_fast, mdq_env_topology on B refined meshes with the workspaces `ws`."""
        sp = C.c_void_p(s.cuda_stream)
        a = state(B)
        args = (B, NV, NT, a["c"].data_ptr(), a["t"].data_ptr(), a["nv"].data_ptr(), a["nt"].data_ptr())
        rcs = [lib.mdq_remesh(*args, a["rem"].data_ptr(), a["st"].data_ptr(), *ws["remesh"], sp),
               lib.mdq_smooth(*args, a["its"].data_ptr(), *ws["smooth"], sp),
               lib.mdq_smooth_fast(*args, a["its"].data_ptr(), *ws["fast"], sp)]
        db = ws["topo_engine"]
        db.coords.copy_(a["c"]); db.cells.copy_(a["t"]); db.nv.copy_(a["nv"]); db.nt.copy_(a["nt"])
        if "topo" in ws:
            db.desc.workspace, db.desc.workspace_bytes = ws["topo"]
        rcs.append(lib.mdq_env_topology(C.byref(db.desc), sp, db.status.data_ptr()))
        return rcs, a

    def workspaces(B):
        eng = DeviceTopologyBatch(B, NV, NT, t0.ne, int((tags == 1).sum()), 180, 1536, polygon, device="cuda", ipcs=False)
        out = dict(topo_engine=eng)
        for k, n in (("remesh", lib.mdq_remesh_workspace_bytes(B, NV, NT)), ("smooth", lib.mdq_smooth_workspace_bytes(B, NV, NT)),
                     ("fast", lib.mdq_smooth_fast_workspace_bytes(B, NV))):
            assert n > 0
            w = torch.empty(int(n), dtype=torch.uint8, device="cuda")
            out[k] = (w.data_ptr(), int(n))
            out["keep_" + k] = w
        return out

    # (1) refused, loudly
    w2 = workspaces(2)
    s0 = torch.cuda.Stream()
    eng = w2["topo_engine"]
    good_topo = (eng.desc.workspace, eng.desc.workspace_bytes)
    assert good_topo[1] == lib.mdq_env_topology_workspace_bytes(C.byref(eng.desc)) > 0
    for bad in (lambda p, n: (None, 0), lambda p, n: (p, n - 1), lambda p, n: (p + 4, n)):
        ws = dict(w2)
        for k in ("remesh", "smooth", "fast"):
            ws[k] = bad(*w2[k])
        ws["topo"] = bad(*good_topo)
        rcs, _ = calls(2, s0, ws)
        assert all(rc < 0 for rc in rcs), rcs
        assert b"workspace" in lib.mdq_last_error() or b"bad arguments" in lib.mdq_last_error()
    # (2) warm-up with the right workspaces, then eight meshes on a new stream: no allocation inside the calls
    w2["topo"] = good_topo
    rcs, a = calls(2, s0, w2)
    assert rcs == [0, 0, 0, 0]
    torch.cuda.synchronize()
    assert (a["st"] == 0).all() and (a["nv"] == NV - 1).all()
    w8, s1 = workspaces(8), s0
    probe = state(8)                                   # (torch's own allocations of calls(): in the cache before the measurement)
    del probe
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rcs, a = calls(8, s1, w8)
    free1 = torch.cuda.mem_get_info()[0]               # NOT synchronised: the calls have returned, the kernels may still run
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    assert rcs == [0, 0, 0, 0]
    assert free1 == free0 and free2 == free0, (free0, free1, free2)
    assert (a["st"] == 0).all() and (a["nv"] == NV - 1).all() and (w8["topo_engine"].status == 0).all()


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: the refined-mesh env step against the ORACLE (global Qhull Delaunay + all-boundary filter = the reference's
# Env2DAirfoil._remove_vertex, Env2DAirfoil.py:452-512; _check_mesh :547-602), not only against the C++ twin: committed
# episodes of tests/golden/oracle_stock_ys930_refined.{json,npz} (make_refined_fixtures.py).  Nothing below runs the oracle.

def _refined_fixture():
    import json
    ep = json.load(open(os.path.join(GOLDEN, "oracle_stock_ys930_refined.json")))
    z = np.load(os.path.join(GOLDEN, "oracle_stock_ys930_refined.npz"))
    return ep, z


def _refined_cfg(ep, z, tmp):
    """An env on the fixture's mesh that reloads the oracle's ground truth (Env2DAirfoil.py:126-153, as tests/test_stock_gpu.py)."""
    tmp = str(tmp)
    path = os.path.join(tmp, "ys930_refined.npz")
    np.savez(path, coords=z["coords"], cells=z["cells"])
    snap = os.path.join(tmp, "snapshots")
    os.makedirs(snap, exist_ok=True)
    u, p = z["u"], z["p"]
    n2 = u.shape[1] // 2
    np.save(os.path.join(snap, "save_velocities.npy"), np.stack([u[:, :n2], u[:, n2:]], axis=2).reshape(len(u), -1))
    np.save(os.path.join(snap, "save_pressures.npy"), p)
    ap = dict(ep["agent_params"])
    ap.update(gt_drag=z["gt_drag"].copy(), gt_lift=z["gt_lift"].copy(), gt_time=np.array([0.05]), plot_dir=tmp)
    return dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=path),
                                 solver_params=dict(dt=0.001, solver_type="lu", smooth=True)), agent_params=ap)


def _refined_script(ep, B=None, seed=0):
    names = list(ep["episodes"])
    B = B or len(names)
    assign = np.arange(B) % len(names) if B == len(names) else np.random.default_rng(seed).permutation(np.arange(B) % len(names))
    K = max(len(ep["episodes"][n]["steps"]) for n in names)
    acts = np.full((K, B), 180, np.int64)
    for b in range(B):
        s = ep["episodes"][names[assign[b]]]["steps"]
        acts[:len(s), b] = [g["action"] for g in s]
    return names, assign, K, acts


def _check_refined(g, where, nv=None, E=None, r=None, done=None, drags=None, lifts=None, cmap=None, removed=None, x_sum=None):
    if done is not None:
        assert bool(done) == g["done"], where
    if r is not None:
        assert abs(r - g["reward"]) < 1e-6, (where, r, g["reward"])
    if nv is not None:
        assert nv == g["nv"], where
    if E is not None:
        assert E == g["E"], where
    if removed is not None:
        assert removed == g["removed_vertex"], where
    if drags is not None:
        assert np.allclose(drags, g["new_drags"], rtol=1e-7, atol=0), where
        assert np.allclose(lifts, g["new_lifts"], rtol=1e-7, atol=0), where
    if cmap is not None:
        assert list(cmap)[:len(g["coord_map"])] == g["coord_map"], where          # all 180 ids of the N-closest window
    if x_sum is not None:
        assert abs(x_sum - g["x_sum"]) < 2e-3, where


@pytest.mark.slow
def test_refined_mesh_episodes_match_the_oracle(lib_built, tmp_path):
    """BASELINE configs[4]'s mesh: `VecEnv2DAirfoil.step` (device mesh engine, host reward logic) and
    `rollout_device(actions=...)` (everything as kernels) against the oracle episodes: removed vertex ids, nv / E and the whole
    coord_map exact at every step, rewards <= 1e-6, interpolated forces <= 1e-7, `done` equal - incl. the episode that removes
    refinement midpoints and ends on the vertex criterion at its 7th removal."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    ep, z = _refined_fixture()
    cfg = _refined_cfg(ep, z, tmp_path)
    names, assign, K, acts = _refined_script(ep)
    base = Env2DAirfoil(cfg)
    assert len(base.flow_solver.mesh.coordinates()) == 3322 and np.array_equal(base.gt_drag, z["gt_drag"])
    venv = VecEnv2DAirfoil(cfg, len(names), base_env=base, auto_reset=False, nthreads=2)
    st = venv.get_state()
    checked = 0
    for k in range(K):
        removed = [int(st["coord_map"][b][acts[k, b]]) if acts[k, b] != 180 else -1 for b in range(len(names))]
        st, rew, done, info = venv.step(acts[k])
        for b, n in enumerate(names):
            s = ep["episodes"][n]["steps"]
            if k < len(s):
                _check_refined(s[k], (n, k), int(info["nv"][b]), int(st["edge_ptr"][b + 1] - st["edge_ptr"][b]), rew[b], done[b],
                               info["new_drags"][b], info["new_lifts"][b], st["coord_map"][b].tolist(), removed[b],
                               float(st["x"][b].double().sum()))
                checked += 1
    assert checked == sum(len(e["steps"]) for e in ep["episodes"].values()) >= 15
    assert any(s["done"] and s["nv"] == 3315 for e in ep["episodes"].values() for s in e["steps"])      # the vertex terminal
    dev = VecEnv2DAirfoil(cfg, len(names), base_env=base, auto_reset=False, nthreads=2)
    dev.get_state()
    for k in range(K):                                  # single steps: every state is read back and compared
        out = dev.rollout_device(None, 1, actions=acts[k:k + 1])
        for b, n in enumerate(names):
            s = ep["episodes"][n]["steps"]
            if k < len(s):
                assert out["codes"][0, b] == 0
                _check_refined(s[k], (n, k, "device"), int(out["nv"][0, b]), int(dev.h["nedges"][b]), out["rewards"][0, b],
                               out["dones"][0, b], dev.new_drags[b], dev.new_lifts[b], dev.h["coord_map"][b].tolist())


@pytest.mark.slow
@pytest.mark.parametrize("flow_steps", [0, 1])
def test_refined_mesh_episodes_at_the_baseline_batch(lib_built, flow_steps, tmp_path):
    """The same episodes dealt out over B = 128 environments (BASELINE configs[4]: 128 refined meshes per GPU; round 4 only
    `bench.py` ran this size, which checks nothing), through `rollout_device` in chunks - the S1 step and the S3 step with the
    IPCS leg on the flow stream: every environment reproduces its episode step for step."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    B = 128
    ep, z = _refined_fixture()
    cfg = _refined_cfg(ep, z, tmp_path)
    names, assign, K, acts = _refined_script(ep, B, seed=128)
    venv = VecEnv2DAirfoil(cfg, B, base_env=Env2DAirfoil(cfg), auto_reset=False, nthreads=4, flow_steps=flow_steps,
                           flow_overlap=bool(flow_steps))
    venv.get_state()
    k0, checked, chunks = 0, 0, [1, 2, 3]
    while k0 < K:
        n = min(chunks.pop(0) if chunks else 4, K - k0)
        out = venv.rollout_device(None, n, actions=acts[k0:k0 + n])
        assert np.isfinite(out["rewards"]).all() and (out["codes"] == 0).all()
        for j in range(n):
            last = j == n - 1
            for b in range(B):
                s = ep["episodes"][names[assign[b]]]["steps"]
                if k0 + j < len(s):
                    _check_refined(s[k0 + j], (names[assign[b]], b, k0 + j), int(out["nv"][j, b]),
                                   int(venv.h["nedges"][b]) if last else None, out["rewards"][j, b], out["dones"][j, b],
                                   venv.new_drags[b] if last else None, venv.new_lifts[b] if last else None,
                                   venv.h["coord_map"][b].tolist() if last else None)
                    checked += 1
        k0 += n
    assert checked == sum(len(ep["episodes"][names[a]]["steps"]) for a in assign)
    if flow_steps:
        fd, fl = venv.flow_wait()
        it = venv.flow_iters.cpu().numpy()
        assert (it[:, 0] > 0).all() and (it[:, 1] > 0).all() and np.isfinite(fd).all() and np.isfinite(fl).all()
        # environments that replayed the same episode hold the same mesh and the same warm start: same re-solved forces
        for a in range(len(names)):
            idx = np.flatnonzero(assign == a)
            assert np.allclose(fd[idx], fd[idx[0]], rtol=1e-9, atol=0) and np.allclose(fl[idx], fl[idx[0]], rtol=1e-9, atol=0)
        # ... and against the sparse-LU oracle on the very same mesh and warm start (as tests/test_stock_gpu.py does on the lab
        # meshes): three sampled environments - the LAST one (its team holds the last block id, 255, on XCD 7), one on an even
        # XCD (66 -> XCD 2) and environment 0.  This is the one place where mode 7's XCD-local team barrier runs with every CU
        # of the chip busy; agreement between environments alone would not see a barrier that fails the same way everywhere.
        from oracle.ipcs import OracleFlowSolver            # (checker)
        for b in (B - 1, 66, 0):
            nv, nt = int(venv.nv[b]), int(venv.nt[b])
            n2 = nv + int(venv.h["ne"][b])
            o = OracleFlowSolver(venv.coords[b, :nv].copy(), venv.cells[b, :nt].copy(), smooth=False)
            assert o.th.np2 == n2
            u0 = venv.u[b, venv.S - 1, :n2].cpu().numpy()
            o.u_n = np.concatenate([u0[:, 0], u0[:, 1]])
            o.p_n = venv.p[b, venv.S - 1, :nv].cpu().numpy().copy()
            _, _, do, lo = o.evolve()
            scale = max(abs(do), abs(lo))       # (1e-7 of the FORCE scale: the leg runs at the Krylov tolerance 1e-10)
            assert abs(fd[b, 0] - do) < 1e-7 * abs(do) and abs(fl[b, 0] - lo) < 1e-7 * scale, (b, fd[b, 0], do, fl[b, 0], lo)


def test_skewed_sweep_schedule_gives_the_bits_of_the_level_schedule(lib_built, tmp_path):
    """The large-mesh smoothing overlaps consecutive sweeps (skew_sweeps: vertex times tau with 1 <= tau(w) - tau(v) <= P - 1
    over the interior edges, found by relaxation per launch): the same updates on the same values as the level-by-level
    kernel (`MDQ_NO_SMOOTH_FLOW=1`, read once per process - hence two child processes) - bit for bit, for 1, 2, 3, 7 and 50
    sweeps (fewer sweeps than overlapping classes included), on the red-refined ys930 and on a coarsened copy of it
    (30 removals: another numbering, another tau)."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(str(tmp_path), "run.py")
    with open(script, "w") as f:
        f.write('''
import hashlib, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from meshdqn_amd.ipcs_batch import smooth_coords
from meshdqn_amd.mesh_ops import red_refine, remesh_batch, smooth_batch_gpu
from meshdqn_amd.topology import MeshTopology
m = np.load(os.path.join(%r, "tests", "golden", "ys930.npz"))
rc, rcells = red_refine(smooth_coords(MeshTopology(m["coords"], m["cells"]), 50), m["cells"])
t = MeshTopology(rc, rcells)
meshes = [(rc.copy(), np.sort(rcells, axis=1).astype(np.int32))]
# a coarsened copy: 30 interior vertices removed through the host engine
c = np.zeros((1, t.nv, 2)); c[0] = rc
tr = np.zeros((1, t.nt, 3), np.int32); tr[0] = meshes[0][1]
nv, nt = np.array([t.nv], np.int32), np.array([t.nt], np.int32)
rng = np.random.default_rng(5)
done = 0
while done < 30:
    bnd = MeshTopology(c[0, :nv[0]], tr[0, :nt[0]]).on_boundary
    v = int(rng.choice(np.flatnonzero(~bnd)))
    st = remesh_batch(c, tr, nv, nt, np.array([v], np.int32))
    done += int(st[0] == 0)
meshes.append((c[0, :nv[0]].copy(), tr[0, :nt[0]].copy()))
h = hashlib.sha256()
for x, cells in meshes:
    for s in (1, 2, 3, 7, 50):
        tc = torch.from_numpy(x[None].copy()).cuda()
        tt = torch.from_numpy(cells[None].copy()).cuda()
        one = lambda v_: torch.full((1,), v_, dtype=torch.int32, device="cuda")
        smooth_batch_gpu(tc, tt, one(len(x)), one(len(cells)), one(s))
        out = tc.cpu().numpy()
        assert np.isfinite(out).all()
        h.update(out.tobytes())
print("DIGEST", h.hexdigest(), len(meshes[1][0]))
''' % (ROOT, ROOT))
    dig = []
    for flow in (True, False):
        env = dict(os.environ)
        env.pop("MDQ_NO_SMOOTH_FLOW", None)
        if not flow:
            env["MDQ_NO_SMOOTH_FLOW"] = "1"
        out = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0].split()
        dig.append(line[1])
        assert int(line[2]) == 3322 - 30
    assert dig[0] == dig[1]


def test_device_built_tile_maps_equal_the_host_built_ones(lib_built):
    """`mdq_ipcs_build_tile_maps` (round 6: the tile maps of the coarsened meshes of the S3 step, built on the device from the
    dof <- slot lists) against what `IpcsBatch` builds on the host for the same cells and dof numbering: row lists (with the
    first / last chunk bits), row counts and packed local maps bit for bit - on the red-refined ys930, on a coarsened copy of it
    (30 removals: a cell order that is no longer the generator's) and with a lab mesh riding along in the big layout; a row
    capacity that is too small marks the environment (-1) instead of writing a truncated list; and an IPCS step through the
    device-built maps alone (mf_scat / mf_tptr NULL) gives the bits of the step through the host-built ones, modes 5 and 7."""
    import ctypes as C
    import torch
    from meshdqn_amd import _lib
    from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
    from meshdqn_amd.mesh_ops import red_refine, remesh_batch
    from meshdqn_amd.topology import MeshTopology
    lib = _lib.load()
    m = np.load(os.path.join(GOLDEN, "ys930.npz"))
    t0 = MeshTopology(m["coords"], m["cells"])
    x0 = smooth_coords(t0, 50)
    rc, rcells = red_refine(x0, m["cells"])
    rt = MeshTopology(rc, rcells)
    c = np.zeros((1, rt.nv, 2)); c[0] = rc
    tr = np.zeros((1, rt.nt, 3), np.int32); tr[0] = np.sort(rcells, axis=1)
    nv, nt = np.array([rt.nv], np.int32), np.array([rt.nt], np.int32)
    rng = np.random.default_rng(11)
    done = 0
    while done < 30:
        bnd = MeshTopology(c[0, :nv[0]], tr[0, :nt[0]]).on_boundary
        st = remesh_batch(c, tr, nv, nt, np.array([int(rng.choice(np.flatnonzero(~bnd)))], np.int32))
        done += int(st[0] == 0)
    ct = MeshTopology(c[0, :nv[0]].copy(), tr[0, :nt[0]].copy())
    topos, xs = [rt, ct, t0, rt], [rc, ct.coords, x0, rc]
    out = {}
    for mode in (5, 7):
        bb = IpcsBatch(topos, xs, rtol=1e-10, mode=mode, pressure_direct=False)
        d = bb.desc
        B, NT, NRL = d.B, d.NT, d.NRL
        NCH = (NT + 1023) // 1024
        assert NRL > 0 and d.rl_flags == 1 and "mf_lpos" in bb.t
        host = {k: bb.t[k].clone() for k in ("mf_rlist", "mf_rcnt", "mf_lpos")}
        dev = {k: torch.full_like(v, 0x5A5A5A5A) for k, v in host.items()}
        d2 = _lib.IpcsDesc()
        C.memmove(C.byref(d2), C.byref(d), C.sizeof(d))
        for k, v in dev.items():
            setattr(d2, k, v.data_ptr())
        status = torch.full((B,), -7, dtype=torch.int32, device="cuda")
        _lib.check(lib.mdq_ipcs_build_tile_maps(C.byref(d2), status.data_ptr(), None), "mdq_ipcs_build_tile_maps")
        torch.cuda.synchronize()
        assert status.cpu().tolist() == [0] * B
        assert torch.equal(dev["mf_rcnt"], host["mf_rcnt"])
        hr, dr, cnt = host["mf_rlist"].cpu().numpy(), dev["mf_rlist"].cpu().numpy(), host["mf_rcnt"].cpu().numpy()
        hl, dl = host["mf_lpos"].cpu().numpy(), dev["mf_lpos"].cpu().numpy()
        for b, t in enumerate(topos):
            for ch in range(NCH):
                assert np.array_equal(hr[b, ch, :cnt[b, ch]], dr[b, ch, :cnt[b, ch]]), (b, ch)
            assert np.array_equal(hl[b, :, :t.nt], dl[b, :, :t.nt]), b
        # a list capacity below what a chunk touches: the environment is marked, the others are built
        small = int(cnt[2].max()) + 1                    # (enough for the lab mesh, not for the refined ones)
        d3 = _lib.IpcsDesc()
        C.memmove(C.byref(d3), C.byref(d2), C.sizeof(d2))
        d3.NRL = small
        rl_small = torch.zeros((B, NCH, small, 2), dtype=torch.int32, device="cuda")
        rc_small = torch.zeros((B, NCH), dtype=torch.int32, device="cuda")
        d3.mf_rlist, d3.mf_rcnt = rl_small.data_ptr(), rc_small.data_ptr()
        _lib.check(lib.mdq_ipcs_build_tile_maps(C.byref(d3), status.data_ptr(), None), "mdq_ipcs_build_tile_maps")
        torch.cuda.synchronize()
        assert status.cpu().tolist() == [1, 1, 0, 1] and rc_small[:, 0].cpu().tolist() == [-1, -1, int(cnt[2, 0]), -1]
        # the step through the device-built maps ALONE == the step through the host-built ones (direct pressure solve from
        # host-built factors: the coarse matrix of the two-level Krylov solve is summed with LDS atomics across waves)
        ref = IpcsBatch(topos, xs, rtol=1e-10, mode=mode, pressure_direct=True)
        dr_, lr_ = ref.evolve(3)
        bb = IpcsBatch(topos, xs, rtol=1e-10, mode=mode, pressure_direct=True)
        for k, v in dev.items():
            setattr(bb.desc, k, v.data_ptr())
        bb.desc.mf_scat = bb.desc.mf_tptr = None
        dd_, ld_ = bb.evolve(3)
        torch.cuda.synchronize()
        assert np.isfinite(dd_.cpu().numpy()).all()
        # (to round-off, not bit for bit: the right-hand side of the direct pressure solve is summed with LDS atomics, and the
        #  descriptor without mf_tptr makes every workgroup read one word more at its start - another arrival order)
        for a_, b_ in ((dd_, dr_), (ld_, lr_), (bb.u_n, ref.u_n), (bb.p_n, ref.p_n)):
            assert (a_ - b_).abs().max().item() <= 1e-11 * b_.abs().max().item(), (mode, (a_ - b_).abs().max().item())
        out[mode] = dd_.cpu().numpy()
        # ... and an environment marked -1 falls back to the dof <- slot path: the same forces to round-off
        bb2 = IpcsBatch(topos, xs, rtol=1e-10, mode=mode, pressure_direct=True)
        for k, v in dev.items():
            setattr(bb2.desc, k, v.data_ptr())
        bb2.desc.mf_scat = bb2.desc.mf_tptr = None
        dev["mf_rcnt"][1, 0] = -1
        df_, _ = bb2.evolve(3)
        torch.cuda.synchronize()
        df_ = df_.cpu().numpy()
        assert np.array_equal(df_[[0, 2, 3]], out[mode][[0, 2, 3]]) and np.abs(df_[1] - out[mode][1]).max() < 1e-9 * np.abs(out[mode][1]).max()


def test_flow_cell_sort_is_a_permutation_along_the_morton_curve(lib_built):
    """`mdq_flow_sort_cells` on the refined fixture mesh (whose cell order is the refinement's: a chunk of 1 024 consecutive
    triangles touches ~5 000 rows) and on a lab mesh in the same batch: the cells come back as a permutation of the input, a
    second array [6][NT] is permuted alike, the Morton keys of the centroids (recomputed here with the same arithmetic) ascend with
    ties in ascending cell id, two runs give the same order - and a chunk then touches fewer rows than the tile maps' lists hold."""
    from meshdqn_amd import _lib
    from meshdqn_amd.topology import MeshTopology
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    lib = _lib.load()
    _, z = _refined_fixture()
    m = np.load(os.path.join(GOLDEN, "ys930.npz"))
    meshes = [(z["coords"], np.sort(z["cells"], axis=1)), (m["coords"], np.sort(m["cells"], axis=1))]
    B, NV, NT = 2, max(len(c) for c, _ in meshes), max(len(t) for _, t in meshes)
    coords = np.zeros((B, NV, 2)); cells = np.zeros((B, NT, 3), np.int32); tag = np.zeros((B, 6, NT), np.int32)
    for b, (c, t) in enumerate(meshes):
        coords[b, :len(c)] = c
        cells[b, :len(t)] = t
        tag[b, :, :len(t)] = np.arange(len(t))[None] * 6 + np.arange(6)[:, None]
    nv = np.array([len(c) for c, _ in meshes], np.int32); nt = np.array([len(t) for _, t in meshes], np.int32)
    outs = []
    for _ in range(2):
        tc, tcells, ttag = (torch.from_numpy(a.copy()).cuda() for a in (coords, cells, tag))
        tnv, tnt = torch.from_numpy(nv).cuda(), torch.from_numpy(nt).cuda()
        _lib.check(lib.mdq_flow_sort_cells(B, NV, NT, tc.data_ptr(), tnv.data_ptr(), tnt.data_ptr(), tcells.data_ptr(),
                                           ttag.data_ptr(), None), "mdq_flow_sort_cells")
        torch.cuda.synchronize()
        outs.append((tcells.cpu().numpy(), ttag.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    sc, st = outs[0]

    def part(x):
        x = x.astype(np.uint64) & 0xFFFF
        for sh, mk in ((8, 0x00FF00FF), (4, 0x0F0F0F0F), (2, 0x33333333), (1, 0x55555555)):
            x = (x | (x << sh)) & mk
        return x
    for b, (c, t) in enumerate(meshes):
        n = len(t)
        src = st[b, 0, :n] // 6                                  # where every sorted cell came from
        assert np.array_equal(np.sort(src), np.arange(n))         # a permutation ...
        assert np.array_equal(sc[b, :n], t[src])                  # ... of the cells, vertex order inside a cell kept ...
        assert np.array_equal(st[b, :, :n], tag[b][:, src])       # ... and of the second array
        assert np.array_equal(sc[b, n:], cells[b, n:])
        lo, hi = c.min(0), c.max(0)
        cen = (c[t[src], :].sum(1)) * (1.0 / 3.0)                  # ((x0 + x1) + x2) / 3 as in the kernel
        q = np.clip(((cen - lo) * (65536.0 / (hi - lo))).astype(np.int64), 0, 65535)
        key = (part(q[:, 0]) | (part(q[:, 1]) << np.uint64(1))).astype(np.uint64)
        assert (np.diff(key.astype(np.int64)) >= 0).all()
        ties = np.diff(key.astype(np.int64)) == 0
        assert (np.diff(src)[ties] > 0).all()
        if n > 2048:        # rows a chunk of 1 024 triangles touches: before and after
            def touched(tr):
                tp = MeshTopology(c, tr)
                return max(len(np.unique(tp.cell_dofs[k:k + 1024])) for k in range(0, n, 1024))
            before, after = touched(t), touched(sc[b, :n])
            assert before > VecEnv2DAirfoil.FLOW_NRL > 3200 > after, (before, after)


def test_twice_refined_mesh_removal_matches_the_host_engine(lib_built, meshes):
    """`mdq_remesh` and `mdq_smooth_fast` beyond 4 096 vertices (round 6: the 16 384-vertex instances - every table incl. the
    131 072-slot edge hash, the positions and the level array of the smoothing on the caller's slab) on ys930 red-refined twice
    (12 924 vertices / 25 120 triangles): 10 consecutive removals on 2 meshes with different action streams (one "do nothing", one
    boundary vertex), smooth(50) after every third one, against the C++ twins, which the CPU suite holds to scipy / Qhull on this
    very mesh: cells as a set, counts, failure codes, coordinates to 1e-12."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import red_refine, remesh_batch, remesh_batch_gpu, smooth_batch_gpu
    from meshdqn_amd.topology import MeshTopology
    coords, cells = meshes["ys930"]
    rc, rcells = red_refine(smooth_coords(MeshTopology(coords, cells), 50), cells)
    rc, rcells = red_refine(rc, rcells)
    t0 = MeshTopology(rc, rcells)
    assert (t0.nv, t0.nt) == (12924, 25120)
    x0 = smooth_coords(t0, 50)
    B, NV, NT = 2, t0.nv, t0.nt
    hc = np.repeat(x0[None], B, 0).copy()
    ht = np.repeat(np.sort(t0.cells, axis=1)[None].astype(np.int32), B, 0).copy()
    hnv = np.full(B, NV, np.int32); hnt = np.full(B, NT, np.int32)
    dc, dtri = torch.from_numpy(hc).cuda(), torch.from_numpy(ht).cuda()
    dnv, dnt = torch.from_numpy(hnv).cuda(), torch.from_numpy(hnt).cuda()
    dst = torch.zeros(B, dtype=torch.int32, device="cuda")
    rng = np.random.default_rng(13)
    interior0 = np.flatnonzero(~t0.on_boundary)
    for step in range(10):
        rem = np.empty(B, np.int32)
        for b in range(B):
            # (an interior vertex of the ORIGINAL numbering that is still interior: the ids shift down by at most `step`)
            t = MeshTopology(hc[b, :hnv[b]], ht[b, :hnt[b]])
            rem[b] = rng.choice(np.flatnonzero(~t.on_boundary))
        if step == 3:
            rem[1] = -1
        if step == 5:
            rem[0] = 0                        # a boundary vertex: refused, mesh untouched
        sweeps = 50 if step % 3 == 0 else 0        # (every third removal is followed by smooth(50), as the env step does)
        hst = remesh_batch(hc, ht, hnv, hnt, rem, sweeps, 2)
        drem = torch.from_numpy(rem).cuda()
        remesh_batch_gpu(dc, dtri, dnv, dnt, drem, dst)
        if sweeps:
            smooth_batch_gpu(dc, dtri, dnv, dnt, torch.where((drem >= 0) & (dst == 0), sweeps, 0).to(torch.int32))
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy() != 0, hst != 0), (step, dst.cpu().numpy(), hst)
        assert np.array_equal(dnv.cpu().numpy(), hnv) and np.array_equal(dnt.cpu().numpy(), hnt)
        gc, gt = dc.cpu().numpy(), dtri.cpu().numpy()
        for b in range(B):
            assert {tuple(r) for r in gt[b, :hnt[b]].tolist()} == {tuple(r) for r in ht[b, :hnt[b]].tolist()}, (step, b)
            assert np.abs(gc[b, :hnv[b]] - hc[b, :hnv[b]]).max() < 1e-12, (step, b)
    assert hnv.tolist() == [NV - 9, NV - 9] and interior0.size > 12000


def _twice_refined(meshes):
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import red_refine
    from meshdqn_amd.topology import MeshTopology
    coords, cells = meshes["ys930"]
    rc, rcells = red_refine(smooth_coords(MeshTopology(coords, cells), 50), cells)
    rc, rcells = red_refine(rc, rcells)
    t = MeshTopology(rc, rcells)
    assert (t.nv, t.nt) == (12924, 25120)
    return t, smooth_coords(t, 50)


@pytest.mark.parametrize("ipcs", [False, True])
def test_twice_refined_mesh_topology_engine_is_bit_identical_to_host_engine(lib_built, meshes, ipcs):
    """`mdq_env_topology` beyond 4 096 vertices (round 6: the 16 384-vertex instance - every table incl. the edge hash and the
    coordinates on the caller's slab, 32-bit owner slots and slot lists) on ys930 red-refined twice (12 924 vertices, 38 043
    edges), after removals, with a shifted window: every output array of the S1 step (edges, dof map, points, airfoil facets,
    removable vertices, N-closest window, state graph) and, with `ipcs`, the index data of the matrix-free IPCS path (flags,
    outflow entries, dof <- slot lists, SELL pattern) against `mdq_env_topology_host`, bit for bit."""
    from meshdqn_amd.mesh_ops import DeviceTopologyBatch, HostTopologyBatch, remesh_batch
    t0, x0 = _twice_refined(meshes)
    tags = t0.facet_tags(x0)
    # (the state polygon: the ORIGINAL airfoil vertices - 182 boundary vertices of ys930 minus the box - as the env passes it)
    coords0, cells0 = meshes["ys930"]
    from meshdqn_amd.topology import MeshTopology
    tb = MeshTopology(coords0, cells0)
    polygon = x0[[v for v in range(tb.nv) if tb.on_boundary[v] and -0.5 < x0[v, 0] < 3 and -0.5 < x0[v, 1] < 0.5]]
    assert 50 < len(polygon) <= 256
    B = 2
    nbo_cap = 2 * (2 * int((tags == 3).sum()) + 1)
    args = (B, t0.nv, t0.nt, t0.ne, int((tags == 1).sum()), 180, 1536, polygon)
    kw = dict(ipcs=True, nbo_cap=nbo_cap) if ipcs else dict(ipcs=False)
    hb = HostTopologyBatch(*args, **kw)
    for b in range(B):
        hb.coords[b], hb.cells[b], hb.nv[b], hb.nt[b] = x0, np.sort(t0.cells, axis=1), t0.nv, t0.nt
    interior = np.flatnonzero(~t0.on_boundary)
    for rnd in range(2):
        rem = np.array([-1 if rnd == 0 else interior[5000], interior[40 + 7 * rnd]], np.int32)
        assert (remesh_batch(hb.coords, hb.cells, hb.nv, hb.nt, rem, 0, 2) == 0).all()
    hb.offset[:] = [0, 3]
    hb.run(2)
    if ipcs:
        kw["nse1_cap"] = hb.NSE1
    db = DeviceTopologyBatch(*args, device="cuda", **kw)
    assert db.workspace is not None and db.workspace.numel() > 2 * 3_000_000
    db.coords.copy_(torch.from_numpy(hb.coords)); db.cells.copy_(torch.from_numpy(hb.cells))
    db.nv.copy_(torch.from_numpy(hb.nv)); db.nt.copy_(torch.from_numpy(hb.nt)); db.offset.copy_(torch.from_numpy(hb.offset))
    db.run()
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in db.t.items()}
    for b in range(B):
        nv, nt, ne = int(hb.nv[b]), int(hb.nt[b]), int(hb.h["ne"][b])
        n2 = nv + ne
        assert ne > 38000
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
    if ipcs:
        gi = {k: v.cpu().numpy() for k, v in db.ti.items()}
        hi = hb.hi
        for b in range(B):
            nv, nt, ne = int(hb.nv[b]), int(hb.nt[b]), int(hb.h["ne"][b])
            n2 = nv + ne
            assert gi["nbo"][b] == hi["nbo"][b] and hi["nbo"][b] > 20
            nbo = int(hi["nbo"][b])
            nbe = int(hi["bo_ptr"][b][nbo])
            checks = dict(cell_outflow=nt, bcu_flag=n2, bcu_gx=n2, bcp_flag=nv, bo_rows=nbo, bo_ptr=nbo + 1, bo_col=nbe,
                          bo_src=nbe, g1_ptr=nv + 1, g1_src=3 * nt, g2_ptr=n2 + 1, g2_src=6 * nt, sl1_off=(nv + 63) // 64 + 1)
            for k, n in checks.items():
                assert np.array_equal(gi[k][b][:n], hi[k][b][:n]), (k, b)
            nse = int(hi["sl1_off"][b][(nv + 63) // 64])
            assert np.array_equal(gi["sl1_col"][b][:nse], hi["sl1_col"][b][:nse])


@pytest.mark.slow
def test_vec_env_steps_the_twice_refined_mesh(lib_built, meshes, tmp_path):
    """The env step past 4 096 vertices (round 6): `VecEnv2DAirfoil` on ys930 red-refined twice (12 924 vertices / 25 120
    triangles - the only member of the family beyond BASELINE configs[4]'s "~8k-tri"): S1 through `step()` and `rollout_device`
    on the device engine (removal, smoothing and topology instances of 16 384 vertices) against the same class on the C++ host
    engine - identical selections / vertex counts / edge counts / terminal flags, rewards and interpolated forces to 1e-9 -, and
    the S3 step: one IPCS step on every coarsened mesh (index data from the large topology instance, `mdq_ipcs_setup_matfree`,
    the element operators with global vectors and global pressure vectors) against the sparse-LU oracle on the very same mesh
    and warm start, 1e-8."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    from oracle.ipcs import OracleFlowSolver
    t0, _ = _twice_refined(meshes)
    path = os.path.join(str(tmp_path), "ys930_refined2.npz")
    np.savez(path, coords=t0.coords, cells=t0.cells)
    cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=path),
                                solver_params=dict(dt=0.001, solver_type="lu", smooth=True, rtol=1e-10)),
               agent_params=dict(solver_steps=5, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1,
                                 gt_time=-1, u=-1, p=-1, time_reward=0.005, save_steps=1, goal_vertices=0.95, plot_dir=""))
    base = Env2DAirfoil(cfg)
    assert len(base.flow_solver.mesh.coordinates()) == 12924
    B, K = 2, 3
    script = np.random.default_rng(23).integers(0, 181, size=(K, B))
    script[1, 1] = 180
    runs = []
    for kw in (dict(gpu_smoothing=False, gpu_topology=False, gpu_remesh=False), dict()):
        env = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2, **kw)
        env.get_state()
        out = []
        for k in range(K):
            st, rew, done, info = env.step(script[k])
            out.append((info["nv"].copy(), st["coord_map"].copy(), st["nedges"].copy(), info["new_drags"].copy(), rew.copy(),
                        done.copy(), st["x"].cpu().numpy()))
        runs.append(out)
    for a, r in zip(runs[1], runs[0]):
        assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1]) and np.array_equal(a[2], r[2])
        assert np.allclose(a[3], r[3], rtol=1e-9, atol=0) and np.allclose(a[4], r[4], rtol=1e-9) and np.array_equal(a[5], r[5])
        assert np.allclose(a[6], r[6], rtol=1e-5, atol=1e-6)
    assert (runs[1][-1][0] < 12924).all()
    dev = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2)
    dev.get_state()
    out = dev.rollout_device(None, K, actions=script)
    for k in range(K):
        assert np.array_equal(out["nv"][k], runs[0][k][0]) and np.array_equal(out["dones"][k], runs[0][k][5])
        assert np.abs(out["rewards"][k] - runs[0][k][4]).max() < 1e-9
    # S3: the flow leg on the coarsened meshes
    venv = VecEnv2DAirfoil(cfg, B, base_env=base, auto_reset=False, nthreads=2, flow_steps=1, flow_rtol=1e-12)
    venv.get_state()
    st, rew, done, info = venv.step(script[0])
    assert (venv.flow_iters.cpu().numpy() > 0).all()
    for b in range(B):
        nv, nt = int(venv.nv[b]), int(venv.nt[b])
        n2 = nv + int(venv.h["ne"][b])
        o = OracleFlowSolver(venv.coords[b, :nv].copy(), venv.cells[b, :nt].copy(), smooth=False)
        assert o.th.np2 == n2
        u0 = venv.u[b, venv.S - 1, :n2].cpu().numpy()
        o.u_n = np.concatenate([u0[:, 0], u0[:, 1]])
        o.p_n = venv.p[b, venv.S - 1, :nv].cpu().numpy().copy()
        uo, po, do, lo = o.evolve()
        assert abs(info["flow_drag"][b, 0] - do) < 1e-8 * abs(do) and abs(info["flow_lift"][b, 0] - lo) < 1e-8 * abs(lo), b


@pytest.mark.slow
def test_twice_refined_mesh_episode_matches_the_oracle(lib_built, tmp_path):
    """ys930 red-refined twice (12 924 vertices) against an ORACLE episode (tests/golden/make_refined2_fixtures.py ->
    oracle_stock_ys930_refined2.{json,npz}: global scipy / Qhull Delaunay + the all-boundary filter, numpy smoothing and
    interpolation, a short oracle ground truth handed over through the snapshot-reload branch): `VecEnv2DAirfoil.step` and
    `rollout_device` - removed vertex ids, nv / E and the whole 180-entry coord_map exact at every step, rewards <= 1e-6,
    interpolated forces <= 1e-7, `done` equal (the episode ends on the vertex criterion at its third removal)."""
    import json
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    ep = json.load(open(os.path.join(GOLDEN, "oracle_stock_ys930_refined2.json")))
    z = np.load(os.path.join(GOLDEN, "oracle_stock_ys930_refined2.npz"))
    cfg = _refined_cfg(ep, z, tmp_path)
    names, assign, K, acts = _refined_script(ep, 2)            # the one episode, in two environments
    base = Env2DAirfoil(cfg)
    assert len(base.flow_solver.mesh.coordinates()) == 12924 and np.array_equal(base.gt_drag, z["gt_drag"])
    steps = ep["episodes"][names[0]]["steps"]
    assert len(steps) >= 3 and steps[-1]["done"] and steps[-1]["nv"] == 12921
    venv = VecEnv2DAirfoil(cfg, 2, base_env=base, auto_reset=False, nthreads=2)
    st = venv.get_state()
    for k in range(K):
        removed = [int(st["coord_map"][b][acts[k, b]]) for b in range(2)]
        st, rew, done, info = venv.step(acts[k])
        for b in range(2):
            _check_refined(steps[k], (k, b), int(info["nv"][b]), int(st["edge_ptr"][b + 1] - st["edge_ptr"][b]), rew[b], done[b],
                           info["new_drags"][b], info["new_lifts"][b], st["coord_map"][b].tolist(), removed[b],
                           float(st["x"][b].double().sum()))
    dev = VecEnv2DAirfoil(cfg, 2, base_env=base, auto_reset=False, nthreads=2)
    dev.get_state()
    for k in range(K):
        out = dev.rollout_device(None, 1, actions=acts[k:k + 1])
        for b in range(2):
            assert out["codes"][0, b] == 0
            _check_refined(steps[k], (k, b, "device"), int(out["nv"][0, b]), int(dev.h["nedges"][b]), out["rewards"][0, b],
                           out["dones"][0, b], dev.new_drags[b], dev.new_lifts[b], dev.h["coord_map"][b].tolist())


@pytest.mark.slow
def test_refined_s3_rollouts_run_whole_episodes(lib_built, tmp_path):
    """180 device-resident S3 steps of 64 refined meshes with auto-reset (threshold 10: episodes run to the vertex criterion at
    the 167th removal and restart in place): every flow leg builds its tile maps on the Morton-sorted private cells (no
    environment marked, under 2 400 touched rows per chunk), the two-workgroup kernel and the on-chip pressure solve never
    report a failed step (`rollout_end` / `flow_wait` raise on a status word or a non-finite force), iteration counts stay in
    their band, and the episodes really end and restart."""
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.gcn_fused import FusedGcn
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    ep, z = _refined_fixture()
    cfg = _refined_cfg(ep, z, tmp_path)
    cfg["agent_params"].update(threshold=10.0, goal_vertices=0.95)
    B = 64
    venv = VecEnv2DAirfoil(cfg, B, base_env=Env2DAirfoil(cfg), flow_steps=1, flow_overlap=True)
    assert venv._flow_tile_maps
    torch.manual_seed(0)
    net = NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    fused = FusedGcn(net.cuda())
    venv.get_state()
    rng = np.random.default_rng(1370)
    ended = 0
    for r0 in range(0, 180, 10):
        ex = np.array([rng.random(B) < 0.5 for _ in range(10)])
        ra = np.array([rng.integers(0, 181, B) for _ in range(10)])
        out = venv.rollout_device(fused, 10, ex, ra)
        fd, fl = venv.flow_wait()
        it = venv.flow_iters.cpu().numpy()
        rc = venv.flow_ts[0]["mf_rcnt"].cpu().numpy()
        assert np.isfinite(out["rewards"]).all() and (out["codes"] == 0).all() and np.isfinite(fd).all() and np.isfinite(fl).all()
        assert (rc[:, 0] > 0).all() and rc.max() < 2400, (r0, rc.min(), rc.max())
        assert (it[:, 0] > 5).all() and (it[:, 0] < 40).all() and (it[:, 1] > 100).all() and (it[:, 1] < 260).all(), (r0, it.max(0))
        ended += int(out["dones"].sum())
    assert ended >= B and int(venv.nv.max()) > 3322 - 20           # every episode ended once and restarted
