"""Host prototype of the PAIRED block step of the blocked smoothing solve (design study for mdq_smooth_linear, round 4).

Single step (round 3):  x_b = M_b g_b,  g_b = sum of the neighbour positions that are not lower-numbered members of block b.
Paired step: for blocks (b, b + 1) split g_{b+1} = e_{b+1} + C x_b  (C: 0/1 coupling of the rows of block b + 1 to the rows
of block b they have as neighbours), so that
      x_b     = M_b g_b
      x_{b+1} = M_{b+1} e_{b+1} + V g_b,        V = M_{b+1} C M_b   (32 x 32, topology only, built once per launch),
i.e. BOTH gathers are issued together and ONE dependent round (gather -> broadcast -> FMA chains -> store) serves 64 rows.
Checks on the lab meshes: the paired form against the sequential sweep, and the number of previous-block references per row.
    python tools/smooth_pair_proto.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from smooth_block_proto import BS, BlockSweep, sweep_sequential, topology  # noqa: E402

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class PairSweep(BlockSweep):
    def __init__(self, nbr, bd):
        super().__init__(nbr, bd)
        rk = {v: i for i, v in enumerate(self.interior)}
        self.early, self.C, self.V, self.maxlate = [], {}, {}, 0
        for r, (v, out) in enumerate(zip(self.interior, self.outside)):
            b = r // BS
            if b % 2 == 1:          # second block of a pair: references to the first one go through C
                late = [w for w in out if not bd[w] and w < v and rk[w] // BS == b - 1]
                self.maxlate = max(self.maxlate, len(late))
                self.early.append([w for w in out if w not in late])
                C = self.C.setdefault(b, np.zeros((BS, BS)))
                for w in late:
                    C[r - b * BS, rk[w] - (b - 1) * BS] += 1.0
            else:
                self.early.append(out)
        for b, C in self.C.items():
            self.V[b] = self.M[b] @ C @ self.M[b - 1]

    def sweep(self, x):
        for b in range(0, self.nb, 2):
            rows0 = self.interior[b * BS:(b + 1) * BS]
            g0 = np.zeros((BS, 2))
            g0[:len(rows0)] = [x[self.early[b * BS + i]].sum(axis=0) for i in range(len(rows0))]
            if b + 1 < self.nb:
                rows1 = self.interior[(b + 1) * BS:(b + 2) * BS]
                e1 = np.zeros((BS, 2))
                e1[:len(rows1)] = [x[self.early[(b + 1) * BS + i]].sum(axis=0) for i in range(len(rows1))]
                x1 = self.M[b + 1] @ e1 + self.V[b + 1] @ g0
                x[rows1] = x1[:len(rows1)]
            x[rows0] = (self.M[b] @ g0)[:len(rows0)]


if __name__ == "__main__":
    for name in ("ys930", "ah93w145"):
        z = np.load(os.path.join(G, f"{name}.npz"))
        coords, cells = z["coords"].astype(float), z["cells"]
        nbr, vcells, bd = topology(cells, len(coords))
        ps = PairSweep(nbr, bd)
        x = coords.copy()
        for s in range(6):                      # past the sweeps with limited steps
            sweep_sequential(x, nbr, vcells, bd)
        xa, xb, xc = x.copy(), x.copy(), x.copy()
        sp = 0
        for s in range(44):
            sp += sweep_sequential(xa, nbr, vcells, bd)
            BlockSweep.sweep(ps, xb)
            ps.sweep(xc)
        print(f"{name}: interior {len(ps.interior)} blocks {ps.nb}; previous-block references per row <= {ps.maxlate}; "
              f"max |V| {max(np.abs(v).max() for v in ps.V.values()):.3f} max |M| {np.abs(ps.M).max():.3f}; "
              f"special updates {sp}; after 44 sweeps |single - sequential| {np.abs(xb - xa).max():.1e} |paired - sequential| {np.abs(xc - xa).max():.1e}")
