"""CPU oracle (TEST INFRASTRUCTURE ONLY).

A plain numpy/scipy fp64 restatement of the reference's hot path
(`flow_solver.py`, `probes.py`, `Env2DAirfoil.py`, `airfoilgcnn.py` of
BaratiLab/MeshDQN) and of the third-party semantics that path relies on
(DOLFIN smoothing / Taylor-Hood IPCS assembly / MUMPS direct solves,
SciPy-Qhull Delaunay, Shapely polygon distance, PyG SAGE/GCN/TopK).

Rules:
  * Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import this package.  The product (`meshdqn_amd/`) never does; it
    fails loudly if its HIP library is missing.
  * Parity pin status: the IPCS solver + probes + smoothing are PINNED by the
    reference's two benchmark CSV rows (tests/golden/kat_rows.json).  Every
    other part (remesh, interpolation, reward, state, Q-networks) is
    "parity unpinned": the reference holds no test, golden vector or fixture
    for it and none of its dependencies (dolfin, torch_geometric, shapely, ray,
    gym) can be imported here, so the restatement follows the documented
    behaviour of those libraries at the cited call sites.
"""
