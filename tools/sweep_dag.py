"""Critical path of the dependency DAG of ALL sweeps of DOLFIN's Gauss-Seidel smoothing (index order, in place):
x_v^(s) needs x_w^(s) of its lower-numbered interior neighbours w < v and x_u^(s-1) of the higher-numbered ones u > v (and its
own x_v^(s-1)); the same edges are the write-after-read constraints of the in-place update.  Prints the depth of one sweep's
DAG, S x that (what a level schedule of one sweep at a time pays) and the depth of the combined DAG of S sweeps with its level
widths (what a wavefront ACROSS sweeps would pay).   python tools/sweep_dag.py [S]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def analyse(name, cells, nv):
    nbr = [set() for _ in range(nv)]
    cnt = {}
    for a, b, c in cells:
        for u, w in ((a, b), (b, c), (a, c)):
            nbr[u].add(w); nbr[w].add(u)
            e = (min(u, w), max(u, w)); cnt[e] = cnt.get(e, 0) + 1
    bnd = np.zeros(nv, bool)
    for (u, w), n in cnt.items():
        if n == 1:
            bnd[u] = bnd[w] = True
    interior = [v for v in range(nv) if not bnd[v]]
    lower = {v: [w for w in nbr[v] if w < v and not bnd[w]] for v in interior}
    upper = {v: [w for w in nbr[v] if w > v and not bnd[w]] for v in interior}
    prev = {v: 0 for v in interior}
    one = None
    widths = {}
    for s in range(S):
        cur = {}
        for v in interior:
            lv = prev[v]
            for w in lower[v]:
                lv = max(lv, cur[w])
            for u in upper[v]:
                lv = max(lv, prev[u])
            cur[v] = lv + 1
            widths[lv + 1] = widths.get(lv + 1, 0) + 1
        if s == 0:
            one = max(cur.values())
        prev = cur
    depth = max(prev.values())
    w = np.array([widths[k] for k in sorted(widths)])
    print(f"{name}: {len(interior)} interior vertices; one sweep {one} levels, {S} sweeps one at a time {S * one}; combined DAG of {S} sweeps: "
          f"{depth} levels ({depth / (S * one):.2f} of that), {depth / S:.1f} per sweep; level width mean {w.mean():.0f} max {w.max()}")


z = np.load(os.path.join(R, "tests/golden/ys930.npz")); analyse("ys930", np.sort(z["cells"], axis=1), len(z["coords"]))
z = np.load(os.path.join(R, "tests/golden/ah93w145.npz")); analyse("ah93w145", np.sort(z["cells"], axis=1), len(z["coords"]))
z = np.load(os.path.join(R, "tests/golden/oracle_stock_ys930_refined.npz")); analyse("ys930 red-refined", z["cells"], len(z["coords"]))
