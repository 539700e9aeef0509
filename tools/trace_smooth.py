"""Critical-path analysis of mdq_smooth from in-kernel timestamps (needs a build with MDQ_CFLAGS=-DMDQ_SMOOTH_TRACE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd import _lib
from meshdqn_amd.mesh_ops import smooth_batch_gpu
from meshdqn_amd.topology import MeshTopology
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ys930.npz"))
t = MeshTopology(z["coords"], z["cells"])
B = 128
c = torch.from_numpy(np.repeat(z["coords"][None], B, 0).copy()).cuda()
cells = torch.from_numpy(np.repeat(np.sort(z["cells"], 1)[None].astype(np.int32), B, 0).copy()).cuda()
nv = torch.full((B,), t.nv, dtype=torch.int32, device="cuda"); nt = torch.full((B,), t.nt, dtype=torch.int32, device="cuda")
it = torch.full((B,), 50, dtype=torch.int32, device="cuda")
lib = _lib.load()
for rep in range(2):
    cc = c.clone(); smooth_batch_gpu(cc, cells, nv, nt, it); torch.cuda.synchronize()
lib.mdq_smooth_trace_host.restype = C.POINTER(C.c_longlong)
ptr = lib.mdq_smooth_trace_host()
tr = np.ctypeslib.as_array(ptr, shape=(64, 1024, 2)).copy()
nbr_ptr, nbr, _, _ = t.vertex_adjacency()
onb = t.on_boundary
t0 = tr[tr > 0].min()
ready, done = tr[..., 0] - t0, tr[..., 1] - t0
print("total cycles", done.max(), "updates", int((tr[..., 1] > 0).sum()))
det, comp, crit = [], [], []
for s in range(50):
    for v in range(t.nv):
        if onb[v]: continue
        dep = 0
        for w in nbr[nbr_ptr[v]:nbr_ptr[v + 1]]:
            if onb[w]: continue
            d = done[s, w] if w < v else (done[s - 1, w] if s > 0 else 0)
            dep = max(dep, d)
        if s > 0: dep = max(dep, done[s - 1, v])
        det.append(ready[s, v] - dep); comp.append(done[s, v] - ready[s, v])
det, comp = np.array(det), np.array(comp)
print("compute (ready -> flag written): mean %.0f  p10 %.0f p50 %.0f p90 %.0f cycles" % (comp.mean(), *np.percentile(comp, [10, 50, 90])))
print("detection (last dependency done -> ready seen): p10 %.0f p50 %.0f p90 %.0f; share with < 400 cycles %.2f" % (*np.percentile(det, [10, 50, 90]), (det < 400).mean()))
# critical path: walk back from the last update through the latest-finishing dependency
s, v = np.unravel_index(np.argmax(done[:50, :t.nv]), (50, t.nv))
n = 0; dsum = 0; csum = 0
while True:
    best, bt = None, -1
    for w in list(nbr[nbr_ptr[v]:nbr_ptr[v + 1]]) + [v]:
        if onb[w]: continue
        if w == v: ss = s - 1
        else: ss = s if w < v else s - 1
        if ss < 0: continue
        if done[ss, w] > bt: bt, best = done[ss, w], (ss, w)
    n += 1; csum += done[s, v] - ready[s, v]
    if best is None: break
    dsum += ready[s, v] - bt
    s, v = best
print("critical path: %d updates, compute %.0f + detection %.0f cycles per update" % (n, csum / n, dsum / n))
