"""Analyse a smooth_lab trace (build with -DMDQ_SMOOTH_TRACE): per-update (ready, done) shader-clock stamps of env 0."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from meshdqn_amd.topology import MeshTopology
mesh, path = sys.argv[1], sys.argv[2]
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 8
z = np.load(os.path.join(ROOT, "tests", "golden", mesh + ".npz"))
t = MeshTopology(z["coords"], z["cells"])
ptr, nbr, _, _ = t.vertex_adjacency()
onb = t.on_boundary
tr = np.fromfile(path, dtype=np.int64).reshape(64, 1024, 2)
S = 50
t0 = tr[tr > 0].min()
ready = tr[:S, :t.nv, 0] - t0
done = tr[:S, :t.nv, 1] - t0
iv = np.nonzero(~onb)[0]
rank = {v: i for i, v in enumerate(iv)}
NG = NW * 8
print("total cycles %d, updates %d, cycles/sweep %.0f" % (done.max(), (tr[..., 1] > 0).sum(), done.max() / S))
comp = (done - ready)[:, iv]
print("pass length (ready -> done): mean %.0f p10 %.0f p50 %.0f p90 %.0f" % (comp.mean(), *np.percentile(comp, [10, 50, 90])))
# per wave: passes = distinct ready stamps
for wv in range(NW):
    vs = [v for v in iv if (rank[v] % NG) % NW == wv]
    ev = sorted(set((ready[s, v], done[s, v]) for s in range(S) for v in vs))
    starts = np.array(sorted(set(e[0] for e in ev)))
    # merge passes: updates with the same ready stamp belong to one pass
    ends = np.array([max(e[1] for e in ev if e[0] == st) for st in starts[:2000]])
    gaps = starts[1:len(ends)] - ends[:-1]
    if wv < 2:
        print("wave %d: %d updates in %d passes; busy %.0f%%; gap p10 %.0f p50 %.0f p90 %.0f" % (
            wv, len(vs) * S, len(starts), 100 * (ends - starts[:len(ends)]).sum() / (ends[-1] - starts[0]), *np.percentile(gaps, [10, 50, 90])))
npass = sum(len(set(ready[s, v] for s in range(S) for v in iv if (rank[v] % NG) % NW == wv)) for wv in range(NW))
print("passes total", npass, "updates per pass %.2f" % (len(iv) * S / npass))
det = []
for s in range(1, S):
    for v in iv:
        dep = done[s - 1, v]
        for w in nbr[ptr[v]:ptr[v + 1]]:
            if onb[w]:
                continue
            dep = max(dep, done[s, w] if w < v else done[s - 1, w])
        det.append(ready[s, v] - dep)
det = np.array(det)
print("detection (last dependency done -> pass start): p10 %.0f p50 %.0f p90 %.0f mean %.0f; share < 300: %.2f" % (*np.percentile(det, [10, 50, 90]), det.mean(), (det < 300).mean()))
# critical path
s, v = np.unravel_index(np.argmax(done), done.shape)
n = 0; dsum = 0; csum = 0; own = 0
while True:
    best, bt = None, -1
    for w in list(nbr[ptr[v]:ptr[v + 1]]) + [v]:
        if onb[w]:
            continue
        ss = s - 1 if w >= v else s
        if ss < 0:
            continue
        if done[ss, w] > bt:
            bt, best = done[ss, w], (ss, w)
    n += 1; csum += done[s, v] - ready[s, v]
    if best is None:
        break
    dsum += ready[s, v] - bt
    own += best[1] == v
    s, v = best
print("critical path: %d updates (%d via the vertex's own previous sweep), pass %.0f + detection %.0f cycles per update" % (n, own, csum / n, dsum / n))
