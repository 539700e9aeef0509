"""Write tools/micro/data/*.bin for the stand-alone kernel labs: the two reference meshes (raw, as loaded) and each
after 20 scripted interior-vertex removals (host engine, smoothed after every removal like the env step)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "micro", "data")
os.makedirs(OUT, exist_ok=True)


def write(name, coords, cells):
    with open(os.path.join(OUT, name + ".bin"), "wb") as f:
        np.array([coords.shape[0], cells.shape[0]], dtype=np.int32).tofile(f)
        np.ascontiguousarray(coords, dtype=np.float64).tofile(f)
        np.ascontiguousarray(np.sort(cells, axis=1), dtype=np.int32).tofile(f)
    print(name, coords.shape, cells.shape)


for name in ("ys930", "ah93w145"):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    write(name, z["coords"], z["cells"])
    from meshdqn_amd.mesh_ops import remesh_batch  # noqa: E402
    from oracle.mesh import OracleMesh  # noqa: E402
    rng = np.random.default_rng(1370)
    coords = z["coords"][None].copy()
    cells = np.sort(z["cells"], axis=1).astype(np.int32)[None].copy()
    nv = np.array([coords.shape[1]], np.int32)
    nt = np.array([cells.shape[1]], np.int32)
    assert remesh_batch(coords, cells, nv, nt, np.array([-1], np.int32), 50)[0] == 0   # smooth only
    for k in range(20):
        onb = OracleMesh(coords[0, :nv[0]], cells[0, :nt[0]]).on_boundary
        cand = np.nonzero(~onb)[0]
        st = remesh_batch(coords, cells, nv, nt, np.array([int(rng.choice(cand))], np.int32), 50)
        assert st[0] == 0
    write(name + "_r20", coords[0, :nv[0]], cells[0, :nt[0]])
