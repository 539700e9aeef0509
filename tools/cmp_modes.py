import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology
z = np.load("tests/golden/ys930.npz")
t = MeshTopology(z["coords"], z["cells"]); x = smooth_coords(t, 50)
res = {}
for mode in (2, 3):
    b = IpcsBatch([t], [x], mode=mode, rtol=1e-12)
    out = []
    for s in range(3):
        d, l = b.evolve(1); torch.cuda.synchronize()
        out.append((b.u_n.cpu().numpy().copy(), b.p_n.cpu().numpy().copy(), d.item(), l.item(), b.iters.cpu().numpy().copy()))
    res[mode] = out
for s in range(3):
    u1, p1, d1, l1, i1 = res[2][s]; u2, p2, d2, l2, i2 = res[3][s]
    print("step", s, "du", np.abs(u1 - u2).max() / np.abs(u1).max(), "dp", np.abs(p1 - p2).max() / np.abs(p1).max(), d1, d2, i1, i2)
    bad = np.argwhere(np.abs(u1 - u2)[0].max(axis=1) > 1e-6 * np.abs(u1).max())[:, 0]
    print("  rows differing:", len(bad), bad[:20], "n2", t.np2, "nv", t.nv)

