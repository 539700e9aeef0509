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

# meshes as the smoothing kernel meets them inside an env step: smoothed, then ONE more vertex removed and not yet
# smoothed (the cavity's neighbours usually take a limited first step: the careful-sweeps-first path of the kernel)
z = np.load(os.path.join(ROOT, "tests", "golden", "ys930.npz"))
rng = np.random.default_rng(5)
coords = z["coords"][None].copy()
cells = np.sort(z["cells"], axis=1).astype(np.int32)[None].copy()
nv = np.array([coords.shape[1]], np.int32)
nt = np.array([cells.shape[1]], np.int32)
remesh_batch(coords, cells, nv, nt, np.array([-1], np.int32), 50)
for k in range(6):
    onb = OracleMesh(coords[0, :nv[0]], cells[0, :nt[0]]).on_boundary
    cand = np.nonzero(~onb)[0]
    assert remesh_batch(coords, cells, nv, nt, np.array([int(rng.choice(cand))], np.int32), 0)[0] == 0
    write("ys930_rm%d" % k, coords[0, :nv[0]], cells[0, :nt[0]])
    remesh_batch(coords, cells, nv, nt, np.array([-1], np.int32), 50)

# ... and one with an interior vertex of more than 8 cells (the exact path inside the speculative sweeps)
while True:
    m = OracleMesh(coords[0, :nv[0]], cells[0, :nt[0]])
    deg = np.array([len(m.vcells[v]) for v in range(m.nv)])
    if (deg[~m.on_boundary] > 8).any():
        break
    cand = np.nonzero(~m.on_boundary)[0]
    assert remesh_batch(coords, cells, nv, nt, np.array([int(rng.choice(cand))], np.int32), 50)[0] == 0
print("interior degrees > 8:", np.nonzero((deg > 8) & ~m.on_boundary)[0], deg[(deg > 8) & ~m.on_boundary])
write("ys930_big", coords[0, :nv[0]], cells[0, :nt[0]])
