"""Host prototype of the BLOCK-INVERSE form of a full-step smoothing sweep (design study for mdq_smooth_linear).

A Gauss-Seidel sweep of DOLFIN's smoothing in which every update is a full step is LINEAR with a matrix that depends on
the topology only:  (D - L) x_new = U x_old + (boundary terms),  D = number of neighbours, L / U = adjacency towards
lower / higher numbered INTERIOR vertices.  Interior ranks are cut into blocks of 32 consecutive ranks; per block
x_B = M_B g_B with M_B = (I - D_B^-1 L_BB)^-1 D_B^-1 (32 x 32 lower triangular, computed once per mesh) and g_B = the sum
of the neighbour positions that are NOT lower-numbered members of the same block (new values for lower blocks, old values
otherwise).  22 dependent block steps per sweep instead of ~119 passes.

This script checks, on the lab meshes, that the block form reproduces the sequential sweep to round-off, and reports how
often a sweep is NOT all-full-steps after the first three (careful) sweeps.

    python tools/smooth_block_proto.py
"""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from chain_depth import load  # noqa: E402

EPS = 3.0e-16
BS = 32


def topology(cells, nv):
    nbr = [set() for _ in range(nv)]
    vcells = [[] for _ in range(nv)]
    cnt = {}
    for t, (a, b, c) in enumerate(cells):
        for k, v in enumerate((a, b, c)):
            o = [int(q) for q in (a, b, c) if q != v]
            vcells[v].append((o[0], o[1]))
        for u, w in ((a, b), (b, c), (a, c)):
            nbr[u].add(int(w))
            nbr[w].add(int(u))
            e = (min(u, w), max(u, w))
            cnt[e] = cnt.get(e, 0) + 1
    bd = np.zeros(nv, bool)
    for (u, w), n in cnt.items():
        if n == 1:
            bd[u] = bd[w] = True
    return [sorted(s) for s in nbr], vcells, bd


def sweep_sequential(x, nbr, vcells, bd):
    """One sweep, DOLFIN semantics; returns the number of limited / skipped updates."""
    special = 0
    for v in range(len(x)):
        if bd[v]:
            continue
        p = x[v].copy()
        c = x[nbr[v]].sum(axis=0) / len(nbr[v])
        rmin = min(abs(((b - a)[1] * (p - a)[0] - (b - a)[0] * (p - a)[1])) / np.hypot(*(b - a)) for a, b in
                   ((x[a_], x[b_]) for a_, b_ in vcells[v]))
        d = c - p
        r = np.hypot(*d)
        if r < EPS:
            special += 1
            continue
        if r > 0.5 * rmin:
            special += 1
            x[v] = p + 0.5 * rmin * d / r
        else:
            x[v] = p + d           # (the kernels add the difference; c itself differs by one rounding)
    return special


class BlockSweep:
    def __init__(self, nbr, bd):
        nv = len(nbr)
        self.interior = [v for v in range(nv) if not bd[v]]
        rk = {v: i for i, v in enumerate(self.interior)}
        n = len(self.interior)
        self.nb = (n + BS - 1) // BS
        self.M = np.zeros((self.nb, BS, BS))
        self.outside = []          # per rank: neighbours that are not in-block lower members
        for b in range(self.nb):
            rows = self.interior[b * BS:(b + 1) * BS]
            m = len(rows)
            N = np.zeros((m, m))
            k = np.array([len(nbr[v]) for v in rows], float)
            for i, v in enumerate(rows):
                out = []
                for w in nbr[v]:
                    if not bd[w] and w < v and rk[w] // BS == b:
                        N[i, rk[w] - b * BS] = 1.0 / k[i]
                    else:
                        out.append(w)
                self.outside.append(out)
            inv = np.linalg.inv(np.eye(m) - N)          # (the kernel: forward substitution per column)
            self.M[b, :m, :m] = inv / k[None, :]
        self.max_out = max(len(o) for o in self.outside)
        self.max_new = max(sum(1 for w in o if not bd[w] and w < v) for v, o in zip(self.interior, self.outside))
        self.max_old = max(sum(1 for w in o if bd[w] or w > v) for v, o in zip(self.interior, self.outside))

    def sweep(self, x):
        for b in range(self.nb):
            rows = self.interior[b * BS:(b + 1) * BS]
            g = np.array([x[self.outside[b * BS + i]].sum(axis=0) for i in range(len(rows))])
            x[rows] = self.M[b, :len(rows), :len(rows)] @ g


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for p in sorted(glob.glob(os.path.join(here, "micro", "data", "*.bin"))):
        if "fan" in p:
            continue
        coords, cells = load(p)
        nbr, vcells, bd = topology(cells, len(coords))
        bs = BlockSweep(nbr, bd)
        x = coords.copy()
        sp = []
        err = 0.0
        for s in range(12):
            y = x.copy()
            sp.append(sweep_sequential(x, nbr, vcells, bd))
            if sp[-1] == 0:
                bs.sweep(y)
                err = max(err, np.abs(y - x).max())
        # error accumulation when the block form runs on its own for 47 sweeps
        xa, xb = x.copy(), x.copy()
        for s in range(47):
            sweep_sequential(xa, nbr, vcells, bd)
            bs.sweep(xb)
        print(f"{os.path.basename(p):18s} interior {len(bs.interior)} blocks {bs.nb} special updates per sweep {sp} "
              f"max slots: outside {bs.max_out} new {bs.max_new} old {bs.max_old}; one sweep |block - sequential| {err:.1e}; "
              f"after 47 sweeps {np.abs(xa - xb).max():.1e}")
