import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology
z = np.load("tests/golden/ys930.npz")
t = MeshTopology(z["coords"], z["cells"]); x = smooth_coords(t, 50)
res = {}
for mode in (1, 2):
    b = IpcsBatch([t], [x], mode=mode, rtol=1e-12)
    out = []
    for s in range(3):
        d, l = b.evolve(1); torch.cuda.synchronize()
        out.append((b.u_n.cpu().numpy().copy(), b.p_n.cpu().numpy().copy(), d.item(), l.item(), b.iters.cpu().numpy().copy()))
    res[mode] = out
for s in range(3):
    u1, p1, d1, l1, i1 = res[1][s]; u2, p2, d2, l2, i2 = res[2][s]
    print("step", s, "du", np.abs(u1 - u2).max() / np.abs(u1).max(), "dp", np.abs(p1 - p2).max() / np.abs(p1).max(), d1, d2, i1, i2)
    bad = np.argwhere(np.abs(u1 - u2)[0].max(axis=1) > 1e-6 * np.abs(u1).max())[:, 0]
    print("  rows differing:", len(bad), bad[:20], "n2", t.np2, "nv", t.nv)

print("---- workspace compare after 1 step from rest")
ws = {}
for mode in (1, 2):
    b = IpcsBatch([t], [x], mode=mode, rtol=1e-12)
    b.evolve(1); torch.cuda.synchronize()
    c = b.cap; NV, NT, NE = c["NV"], c["NT"], c["NE"]; N2 = NV + NE
    w = b.t["work"].cpu().numpy()
    ws[mode] = (w[12 * NT:12 * NT + 2 * N2].reshape(N2, 2).copy(), w[12 * NT + 12 * N2:12 * NT + 12 * N2 + NV].copy(), b.u_n.cpu().numpy()[0].copy())
us1, pn1, un1 = ws[1]; us2, pn2, un2 = ws[2]
sd = None
print("xs (mode1 holds S*u_new, mode2 holds u*): skip; pnew diff", np.abs(pn1 - pn2).max() / np.abs(pn1).max())
print("u_n diff", np.abs(un1 - un2).max(), "max", np.abs(un1).max())
k = np.argmax(np.abs(un1 - un2).max(axis=1)); print("worst row", k, un1[k], un2[k], "is vertex" if k < t.nv else "edge")
ratio = un2[:, 0] / np.where(np.abs(un1[:, 0]) > 1e-12, un1[:, 0], np.nan)
print("ratio stats", np.nanmin(ratio), np.nanmax(ratio), np.nanmedian(ratio))
bc = t.boundary_conditions(x)
fl = bc["bcu_flag"].astype(bool); gx = bc["bcu_gx"]
rows = np.flatnonzero(fl & (gx != 0))
print("inlet rows:", rows, "k idx", rows // 512)
print("mode1 vals", un1[rows, 0]); print("mode2 vals", un2[rows, 0]); print("g", gx[rows])
