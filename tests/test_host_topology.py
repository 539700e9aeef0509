"""CPU tests: product host logic (topology, BCs, smoothing) against the oracle."""
import numpy as np
import pytest

from meshdqn_amd.topology import MeshTopology
from oracle.mesh import OracleMesh
from oracle.fem import TaylorHood


@pytest.mark.parametrize("name,nv,nt,ne,nb", [("ys930", 876, 1570, 2446, 182), ("ah93w145", 797, 1431, 2228, 163)])
def test_topology_counts_and_oracle_agreement(meshes, name, nv, nt, ne, nb):
    coords, cells = meshes[name]
    t = MeshTopology(coords, cells)
    o = OracleMesh(coords, cells)
    assert (t.nv, t.nt, t.ne) == (nv, nt, ne)
    assert int(t.on_boundary.sum()) == nb == len(t.boundary_edges)
    # identical edge numbering (first appearance) and cell->edge maps
    assert np.array_equal(t.edges, o.edges)
    assert np.array_equal(t.cell_edges, o.cell_edges)
    assert np.array_equal(t.on_boundary, o.on_boundary)
    assert np.array_equal(t.removable(), o.removable())
    # Euler characteristic of a domain with one hole
    assert t.nv - t.ne + t.nt == 0


def test_facet_tags_and_bcs(meshes):
    coords, cells = meshes["ys930"]
    t = MeshTopology(coords, cells)
    o = OracleMesh(coords, cells)
    tags = t.facet_tags()
    otags = o.facet_tags()
    assert [otags[int(e)] for e in t.boundary_edges] == list(tags)
    assert np.bincount(tags, minlength=5).tolist() == [48, 120, 7, 7, 0]
    th = TaylorHood(o)
    bc = t.boundary_conditions()
    sd = np.flatnonzero(bc["bcu_flag"])
    assert np.array_equal(sd, th.bcu_scalar_dofs)
    assert 2 * len(sd) == 702
    assert np.allclose(bc["bcu_gx"][sd], th.bcu_vals[:len(sd)], rtol=0, atol=0)
    assert np.array_equal(np.flatnonzero(bc["bcp_flag"]), th.bcp_dofs)
    assert len(th.bcp_dofs) == 8


def test_patterns_match_oracle(meshes):
    coords, cells = meshes["ah93w145"]
    t = MeshTopology(coords, cells)
    o = OracleMesh(coords, cells)
    th = TaylorHood(o)
    pat = t.patterns()
    rp, ci, ap, asrc = pat["p2"]
    M = th.M.tocsr()
    M.sort_indices()
    assert rp[-1] == 33565
    # structural pattern of cell-dof couplings equals the assembled oracle mass matrix pattern
    # (mass entries can cancel to exact zero only for vertex/opposite-edge pairs, which scipy keeps)
    assert np.array_equal(rp, M.indptr)
    assert np.array_equal(ci, M.indices)
    # every element slot appears exactly once in the gather map
    assert np.array_equal(np.sort(asrc), np.arange(36 * t.nt))
    rp1, ci1, ap1, asrc1 = pat["p1"]
    assert rp1[-1] == 5253
    g = t.dof_gathers()
    ptr, src = g["p2"]
    assert np.array_equal(t.cell_dofs.ravel()[src], np.repeat(np.arange(t.np2), np.diff(ptr)))


def test_host_smoothing_matches_oracle(meshes, lib_built):
    from meshdqn_amd.ipcs_batch import smooth_coords
    coords, cells = meshes["ys930"]
    t = MeshTopology(coords, cells)
    x = smooth_coords(t, 50)
    o = OracleMesh(coords, cells).smooth(50)
    assert np.abs(x - o.coords).max() < 1e-13
    # boundary untouched, interior moved (displacements are a few percent of the chord)
    assert np.array_equal(x[t.on_boundary], coords[t.on_boundary])
    disp = np.linalg.norm(x - coords, axis=1)
    assert 0.02 < disp.max() < 0.05


def test_library_exports_exactly_the_declared_symbols(lib_built):
    """the dynamic symbol table of the library == the MDQ_API declarations of the header == the ctypes binding: no kernel
    stub, no C++ helper, no libstdc++ instantiation leaks out (-fvisibility=hidden + the linker version script)."""
    import re, os, subprocess
    from meshdqn_amd import _lib, build
    lib = _lib.load()
    hdr = open(os.path.join(os.path.dirname(_lib.HERE), "include", "meshdqn_hip.h")).read()
    declared = set(re.findall(r"\b(mdq_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(build.declared_symbols())          # every declaration carries MDQ_API
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mdq_abi_version() == _lib.ABI_VERSION
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_built]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)


def test_pressure_direct_factors_reproduce_dense_solve(meshes):
    """Substructuring factors (host setup of the direct pressure solver) against numpy's solve."""
    from meshdqn_amd.pressure_direct import build_pressure_direct, solve_reference
    from oracle.ipcs import OracleFlowSolver
    coords, cells = meshes["ah93w145"]
    fs = OracleFlowSolver(coords, cells, smooth=False, factorize=False)
    K = fs.A2.toarray()
    sd = np.sqrt(np.diag(K))
    Kt = K / np.outer(sd, sd)
    for parts in (1, 4, 16):
        pd = build_pressure_direct(fs.mesh.coords, Kt, parts)
        assert pd["nI"] + pd["nG"] == K.shape[0]
        assert sorted(pd["node"].tolist()) == list(range(K.shape[0]))
        rng = np.random.default_rng(parts)
        b = rng.standard_normal(K.shape[0])
        x = solve_reference(pd, b)
        xr = np.linalg.solve(Kt, b)
        assert np.abs(x - xr).max() / np.abs(xr).max() < 1e-11


@pytest.mark.parametrize("cname,pyname", [("mdq_ipcs_desc", "IpcsDesc"), ("mdq_interp_desc", "InterpDesc"),
                                          ("mdq_env_topo_desc", "EnvTopoDesc"), ("mdq_ipcs_topo_out", "IpcsTopoOut"),
                                          ("mdq_env_finish_desc", "EnvFinishDesc")])
def test_descriptor_layout_matches_c_header(tmp_path, cname, pyname):
    """every ctypes mirror has the same size / field offsets as its struct in the C header."""
    import ctypes as C, os, subprocess
    from meshdqn_amd import _lib
    cls = getattr(_lib, pyname)
    hdr = os.path.join(os.path.dirname(_lib.HERE), "include", "meshdqn_hip.h")
    fields = [n for n, _ in cls._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{hdr}"', 'int main(){',
           f'printf("%zu\\n", sizeof({cname}));']
    src += [f'printf("%zu\\n", offsetof({cname}, {f}));' for f in fields]
    src.append('return 0;}')
    cfile = tmp_path / "lay.c"
    cfile.write_text("\n".join(src))
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", str(cfile), "-o", str(exe)])
    vals = [int(x) for x in subprocess.check_output([str(exe)]).decode().split()]
    assert vals[0] == C.sizeof(cls)
    assert vals[1:] == [getattr(cls, f).offset for f in fields]


@pytest.mark.parametrize("name", ["ys930", "ah93w145"])
def test_host_engine_topology_matches_python_topology(meshes, lib_built, name):
    """C++ batched engine (mdq_env_topology_host incl. the matrix-free IPCS index data) against MeshTopology /
    IpcsBatch._host_arrays on the smoothed mesh and on a mesh with one vertex removed."""
    from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
    from meshdqn_amd.mesh_ops import HostTopologyBatch, remesh_batch
    coords, cells = meshes[name]
    t0 = MeshTopology(coords, cells)
    x0 = smooth_coords(t0, 50)
    tags = t0.facet_tags(x0)
    polygon = x0[[v for v in range(t0.nv) if t0.on_boundary[v] and -0.5 < x0[v, 0] < 3 and -0.5 < x0[v, 1] < 0.5]]
    B = 2
    hb = HostTopologyBatch(B, t0.nv, t0.nt, t0.ne, int((tags == 1).sum()), 180, 1536, polygon, ipcs=True)
    for b in range(B):
        hb.coords[b], hb.cells[b], hb.nv[b], hb.nt[b] = x0, np.sort(cells, axis=1), t0.nv, t0.nt
    interior = np.flatnonzero(~t0.on_boundary)
    status = remesh_batch(hb.coords, hb.cells, hb.nv, hb.nt, np.array([-1, interior[40]], np.int32), 50, 2)
    assert (status == 0).all() and hb.nv.tolist() == [t0.nv, t0.nv - 1]
    hb.run(2)
    for b in range(B):
        nv, nt = int(hb.nv[b]), int(hb.nt[b])
        t = MeshTopology(hb.coords[b, :nv], hb.cells[b, :nt])
        ref = IpcsBatch._host_arrays(t, hb.coords[b, :nv])
        ne, n2 = t.ne, t.np2
        assert hb.h["ne"][b] == ne
        assert np.array_equal(hb.h["cell_dofs"][b][:, :nt], ref["cell_dofs_soa"])
        # N-closest selection: polygon distances (numpy restatement), argsort, window of 180 (Env2DAirfoil.py:293-315)
        from meshdqn_amd.mesh_ops import polygon_distance
        removable = np.flatnonzero(t.removable(hb.coords[b, :nv]))
        assert hb.h["nremovable"][b] == removable.size
        order = np.argsort(polygon_distance(polygon, hb.coords[b, :nv][removable]), kind="stable")[:180]
        assert np.array_equal(hb.h["n_closest"][b], order)
        assert np.array_equal(hb.h["coord_map"][b], removable[order])
        hi = hb.hi
        assert np.array_equal(hi["cell_outflow"][b][:nt], ref["cell_outflow"])
        assert np.array_equal(hi["bcu_flag"][b][:n2], ref["bcu_flag"])
        assert np.array_equal(hi["bcu_gx"][b][:n2], ref["bcu_gx"])       # same formula, same operation order
        assert np.array_equal(hi["bcp_flag"][b][:nv], ref["bcp_flag"])
        nbo = int(hi["nbo"][b])
        assert nbo == ref["bo_rows"].size
        assert np.array_equal(hi["bo_rows"][b][:nbo], ref["bo_rows"])
        assert np.array_equal(hi["bo_ptr"][b][:nbo + 1], ref["bo_ptr"])
        nbe = int(ref["bo_ptr"][-1])
        assert np.array_equal(hi["bo_col"][b][:nbe], ref["bo_col"]) and np.array_equal(hi["bo_src"][b][:nbe], ref["bo_src"])
        assert np.array_equal(hi["mf_scat"][b][:, :nt] & 0xFFF, ref["mf_scat"] & 0xFFF)
        assert np.array_equal((hi["mf_scat"][b][0, :nt] >> 28) & 3, (ref["mf_scat"][0] >> 28) & 3)
        assert np.array_equal(hi["g1_ptr"][b][:nv + 1], ref["g1_ptr"]) and np.array_equal(hi["g1_src"][b][:3 * nt], ref["g1_src"])
        assert np.array_equal(hi["g2_ptr"][b][:n2 + 1], ref["g2_ptr"]) and np.array_equal(hi["g2_src"][b][:6 * nt], ref["g2_src"])
        ns = (nv + 63) // 64
        assert np.array_equal(hi["sl1_off"][b][:ns + 1], ref["sl1_off"])
        assert np.array_equal(hi["sl1_col"][b][:ref["sl1_col"].size], ref["sl1_col"])


def _scipy_remove(coords, cells, idx):
    """Reference semantics of Env2DAirfoil._remove_vertex (Env2DAirfoil.py:452-512): delete the vertex, global
    scipy/Qhull Delaunay of the remaining points, drop simplices whose three vertices are all boundary vertices."""
    from scipy.spatial import Delaunay
    t = MeshTopology(coords, cells)
    bnd = np.flatnonzero(t.on_boundary)
    bnd = np.where(bnd > idx, bnd - 1, bnd)
    keep = np.delete(np.arange(len(coords)), idx)
    x = coords[keep]
    tri = Delaunay(x).simplices
    tri = tri[np.isin(tri, bnd).sum(axis=1) != 3]
    return x, np.sort(tri, axis=1)


@pytest.mark.parametrize("name", ["ys930", "ah93w145"])
def test_host_remesh_equals_scipy_delaunay_over_an_episode(meshes, lib_built, name):
    """C++ engine (star re-triangulation of the cavity + Lawson flips, then DOLFIN smoothing) against the reference's
    own pipeline (global Qhull Delaunay + all-boundary filter) + the oracle smoother, 24 consecutive removals."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import remesh_batch
    coords, cells = meshes[name]
    t0 = MeshTopology(coords, cells)
    x = smooth_coords(t0, 50)
    c = np.sort(cells, axis=1)
    NV, NT = t0.nv, t0.nt
    hc = np.zeros((1, NV, 2)); hc[0] = x
    ht = np.zeros((1, NT, 3), np.int32); ht[0] = c
    nv = np.array([NV], np.int32); nt = np.array([NT], np.int32)
    rng = np.random.default_rng(3)
    ref_x, ref_c = x.copy(), c.copy()
    for step in range(24):
        t = MeshTopology(ref_x, ref_c)
        interior = np.flatnonzero(~t.on_boundary)
        idx = int(rng.choice(interior))
        st = remesh_batch(hc, ht, nv, nt, np.array([idx], np.int32), 50, 1)
        assert st[0] == 0
        ref_x, ref_c = _scipy_remove(ref_x, ref_c, idx)
        assert (nv[0], nt[0]) == (len(ref_x), len(ref_c))
        mine = {tuple(r) for r in ht[0, :nt[0]].tolist()}
        assert mine == {tuple(r) for r in ref_c.tolist()}, step            # same triangulation as a SET of cells
        ref_x = smooth_coords(MeshTopology(ref_x, ref_c), 50)               # host smoother (pinned to the oracle above)
        assert np.abs(hc[0, :nv[0]] - ref_x).max() < 1e-13
    # (the host smoother itself is pinned to the python oracle by test_host_smoothing_matches_oracle)


@pytest.mark.parametrize("smoothed", [True, False])
def test_host_remesh_equals_scipy_delaunay_on_the_red_refined_mesh(meshes, lib_built, smoothed):
    """BASELINE configs[4] (ys930 red-refined once: 3 322 vertices): the C++ engine against the reference's pipeline
    (`Env2DAirfoil._remove_vertex`, Env2DAirfoil.py:452-512: global Qhull Delaunay + all-boundary filter) over 48 consecutive
    removals - refinement MIDPOINTS (ids >= 876: every one sits in the middle of a parent edge, between two pairs of
    congruent sub-triangles) and original vertices alternately, every third pick among the 60 interior vertices nearest to the
    airfoil, where the policy acts.  Set equality of the cells at every step, on the smoothed mesh (what the env steps, with
    `smooth(50)` after every removal as FlowSolver.remesh does, flow_solver.py:236-237) and on the raw refinement, whose
    parallelogram pairs are the nearest thing to co-circular quads this mesh family has (no disagreement found: the local
    cavity re-triangulation + Lawson flips reproduce Qhull on it)."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import red_refine, remesh_batch
    coords, cells = meshes["ys930"]
    rc, rcells = red_refine(smooth_coords(MeshTopology(coords, cells), 50), cells)
    t0 = MeshTopology(rc, rcells)
    assert (t0.nv, t0.nt) == (3322, 6280)
    x = smooth_coords(t0, 50) if smoothed else rc.copy()
    c = np.sort(rcells, axis=1).astype(np.int32)
    hc = np.zeros((1, t0.nv, 2)); hc[0] = x
    ht = np.zeros((1, t0.nt, 3), np.int32); ht[0] = c
    nv = np.array([t0.nv], np.int32); nt = np.array([t0.nt], np.int32)
    rng = np.random.default_rng(5)
    ref_x, ref_c = x.copy(), c.copy()
    n_mid = 0
    for step in range(48):
        t = MeshTopology(ref_x, ref_c)
        interior = np.flatnonzero(~t.on_boundary)
        n_orig = 876 - (step - n_mid)                  # original vertices still present (ids below this)
        if step % 3 == 2:                              # near the airfoil
            af = np.flatnonzero(t.on_boundary & (np.abs(ref_x[:, 1]) < 0.45) & (ref_x[:, 0] > -0.45) & (ref_x[:, 0] < 2.9))
            d = np.min(np.linalg.norm(ref_x[interior][:, None] - ref_x[af][None], axis=2), axis=1)
            cand = interior[np.argsort(d)[:60]]
        elif step % 2 == 0:
            cand = interior[interior >= n_orig]
        else:
            cand = interior[interior < n_orig]
        idx = int(rng.choice(cand))
        n_mid += idx >= n_orig
        st = remesh_batch(hc, ht, nv, nt, np.array([idx], np.int32), 50 if smoothed else 0, 1)
        assert st[0] == 0
        ref_x, ref_c = _scipy_remove(ref_x, ref_c, idx)
        assert (nv[0], nt[0]) == (len(ref_x), len(ref_c))
        assert {tuple(r) for r in ht[0, :nt[0]].tolist()} == {tuple(r) for r in ref_c.tolist()}, (step, idx)
        if smoothed:
            ref_x = smooth_coords(MeshTopology(ref_x, ref_c), 50)
            assert np.abs(hc[0, :nv[0]] - ref_x).max() < 1e-13
    assert n_mid >= 16


def test_host_remesh_equals_scipy_delaunay_on_the_twice_refined_mesh(meshes, lib_built):
    """ys930 red-refined TWICE (12 924 vertices / 25 120 triangles: the only member of the family beyond BASELINE configs[4]'s
    "~8k-tri"): the C++ engine - the twin `mdq_remesh`'s 16 384-vertex instance is tested against on the GPU - equals the
    reference's global Qhull Delaunay + all-boundary filter (Env2DAirfoil.py:452-512) over 24 consecutive removals: vertices of
    all three generations (original, first and second refinement), every third pick near the airfoil; set equality of the cells."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import red_refine, remesh_batch
    coords, cells = meshes["ys930"]
    rc, rcells = red_refine(smooth_coords(MeshTopology(coords, cells), 50), cells)
    rc, rcells = red_refine(rc, rcells)
    t0 = MeshTopology(rc, rcells)
    assert (t0.nv, t0.nt) == (12924, 25120)
    x = smooth_coords(t0, 50)
    c = np.sort(rcells, axis=1).astype(np.int32)
    hc = np.zeros((1, t0.nv, 2)); hc[0] = x
    ht = np.zeros((1, t0.nt, 3), np.int32); ht[0] = c
    nv = np.array([t0.nv], np.int32); nt = np.array([t0.nt], np.int32)
    rng = np.random.default_rng(9)
    ref_x, ref_c = x.copy(), c.copy()
    picked = []
    for step in range(24):
        t = MeshTopology(ref_x, ref_c)
        interior = np.flatnonzero(~t.on_boundary)
        if step % 3 == 2:                              # near the airfoil
            af = np.flatnonzero(t.on_boundary & (np.abs(ref_x[:, 1]) < 0.45) & (ref_x[:, 0] > -0.45) & (ref_x[:, 0] < 2.9))
            d = np.min(np.linalg.norm(ref_x[interior][:, None] - ref_x[af][None], axis=2), axis=1)
            cand = interior[np.argsort(d)[:120]]
        else:                                          # one of the three generations in turn (ids shift down as vertices go)
            lo, hi = [(0, 800), (900, 3200), (3400, len(ref_x))][step % 3 if step % 3 < 2 else 0] if step % 2 else (3400, len(ref_x))
            cand = interior[(interior >= lo) & (interior < hi)]
        idx = int(rng.choice(cand))
        picked.append(idx)
        st = remesh_batch(hc, ht, nv, nt, np.array([idx], np.int32), 0, 1)
        assert st[0] == 0
        ref_x, ref_c = _scipy_remove(ref_x, ref_c, idx)
        assert (nv[0], nt[0]) == (len(ref_x), len(ref_c))
        assert {tuple(r) for r in ht[0, :nt[0]].tolist()} == {tuple(r) for r in ref_c.tolist()}, (step, idx)
        assert np.array_equal(hc[0, :nv[0]], ref_x)
    assert min(picked) < 900 and max(picked) > 3400


def test_host_engine_is_thread_safe(meshes, lib_built):
    """Concurrent callers (the env groups of VecEnvGroups) share one worker pool: results equal the sequential ones."""
    import threading
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import remesh_batch
    coords, cells = meshes["ys930"]
    t0 = MeshTopology(coords, cells)
    x = smooth_coords(t0, 50)
    c = np.sort(cells, axis=1).astype(np.int32)
    interior = np.flatnonzero(~t0.on_boundary)
    B, G = 6, 4

    def run(g, out):
        hc = np.repeat(x[None], B, 0).copy()
        ht = np.repeat(c[None], B, 0).copy()
        nv = np.full(B, t0.nv, np.int32); nt = np.full(B, t0.nt, np.int32)
        for k in range(3):
            rem = interior[(np.arange(B) * 17 + 31 * g + 5 * k) % (interior.size - 8)].astype(np.int32)
            st = remesh_batch(hc, ht, nv, nt, rem, 10, 3)
            assert (st == 0).all()
        out[g] = (hc, ht, nv.copy(), nt.copy())

    seq, par = {}, {}
    for g in range(G):
        run(g, seq)
    threads = [threading.Thread(target=run, args=(g, par)) for g in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for g in range(G):
        for a, b in zip(seq[g], par[g]):
            assert np.array_equal(a, b)


def test_morton_cell_order_makes_chunks_share_their_rows(meshes):
    """`topology.morton_cell_order` (the host twin of `mdq_flow_sort_cells`; what `IpcsBatch(cell_order="auto")` applies to meshes
    beyond the LDS-resident operator modes): a permutation of the cells, ascending Morton keys with ties in ascending cell id, and
    on the red-refined ys930 a chunk of 1 024 consecutive triangles touches ~2 200 P2 dofs instead of ~5 000 in the refinement's
    own order or in the conflict-free order of the LDS-atomic mode (the kernels' LDS stage of a chunk's input rows holds 4 064)."""
    from meshdqn_amd.ipcs_batch import smooth_coords
    from meshdqn_amd.mesh_ops import red_refine
    from meshdqn_amd.topology import conflict_free_cell_order, morton_cell_order
    coords, cells = meshes["ys930"]
    rc, rcells = red_refine(smooth_coords(MeshTopology(coords, cells), 50), cells)
    perm = morton_cell_order(rc, rcells)
    assert np.array_equal(np.sort(perm), np.arange(len(rcells)))
    assert np.array_equal(perm, morton_cell_order(rc, rcells))

    def touched(cs):
        t = MeshTopology(rc, cs)
        return [len(np.unique(t.cell_dofs[k:k + 1024])) for k in range(0, len(cs), 1024)]
    own, cf, mo = touched(rcells), touched(rcells[conflict_free_cell_order(MeshTopology(rc, rcells).cells)]), touched(rcells[perm])
    assert max(mo) < 2400 < 4064 < min(max(own), max(cf)), (own, cf, mo)
    assert sum(mo) < 1.1 * MeshTopology(rc, rcells).np2 and sum(own) > 2.2 * MeshTopology(rc, rcells).np2
    # a mesh with coincident centroid keys: ties keep the cell order (stable)
    tiny = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    assert morton_cell_order(tiny, np.array([[0, 1, 2], [0, 1, 2], [1, 3, 2]])).tolist() in ([0, 1, 2], [2, 0, 1])
