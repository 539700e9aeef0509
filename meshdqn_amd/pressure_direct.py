"""Host-side setup of the direct (substructured) pressure solver.

The reference solves the pressure Poisson system with a sparse direct solver,
factorised once per mesh (`LUSolver("mumps")`, `flow_solver.py:150-159`).  The
MI355X counterpart is a one-level substructuring factorisation whose solve
phase is nothing but dense, coalesced matrix-vector products (three workgroup
barriers per solve instead of ~150 Krylov iterations):

    nodes = interiors I_0..I_{k-1} of k subdomains  +  vertex separator G
    W_s   = inv(K[I_s, I_s])                 dense, per subdomain
    F_s   = W_s K[I_s, G_s]                  dense, G_s = separator nodes touching I_s
    S     = K[G,G] - sum_s K[G,I_s] W_s K[I_s,G]   (Schur complement),  Sinv = inv(S)

    solve K x = b:   y_I = W b_I ;  g = b_G - K[G,I] y_I ;  x_G = Sinv g ;  x_I = y_I - F x_G

Setup is plain numpy on the host (once per mesh, like the reference's
factorisation); the solve runs inside the HIP time-stepping kernel.
"""
from __future__ import annotations

import numpy as np


def rcb_partition(coords: np.ndarray, nparts: int) -> np.ndarray:
    """Recursive coordinate bisection into `nparts` (power of two) balanced parts."""
    n = coords.shape[0]
    part = np.zeros(n, dtype=np.int32)
    groups = [np.arange(n)]
    while len(groups) < nparts:
        new = []
        for g in groups:
            if g.size <= 1:
                new.extend([g, g[:0]])
                continue
            ext = coords[g].max(axis=0) - coords[g].min(axis=0)
            ax = int(np.argmax(ext))
            order = g[np.argsort(coords[g, ax], kind="stable")]
            h = order.size // 2
            new.extend([order[:h], order[h:]])
        groups = new
    for i, g in enumerate(groups):
        part[g] = i
    return part


def build_pressure_direct(coords: np.ndarray, K: np.ndarray, nparts: int = 16) -> dict:
    """Substructuring factors of the dense SPD matrix K (n,n) with sparsity from a mesh.

    Returns a dict of flat arrays (see the field comments in include/meshdqn_hip.h, `pd_*`).
    """
    n = K.shape[0]
    nparts = int(min(nparts, max(1, 2 ** int(np.floor(np.log2(max(n // 8, 1)))))))
    part = rcb_partition(coords[:n], nparts)
    # vertex separator: walk the off-diagonal couplings; when both ends are still interior and lie in
    # different parts, move the end in the higher-numbered part to the separator
    in_sep = np.zeros(n, dtype=bool)
    ii, jj = np.nonzero(np.triu(K != 0.0, 1))
    for a, b in zip(ii.tolist(), jj.tolist()):
        if part[a] != part[b] and not in_sep[a] and not in_sep[b]:
            in_sep[b if part[b] > part[a] else a] = True
    interiors = [np.flatnonzero((part == s) & ~in_sep) for s in range(nparts)]
    interiors = [g for g in interiors if g.size > 0]
    sep = np.flatnonzero(in_sep)
    nI = int(sum(g.size for g in interiors))
    nG = int(sep.size)
    node = np.concatenate(interiors + [sep]).astype(np.int32)
    inv = np.empty(n, dtype=np.int64)
    inv[node] = np.arange(n)
    meta = np.zeros((len(interiors), 6), dtype=np.int32)
    rowblk = np.zeros(max(nI, 1), dtype=np.int32)
    W_parts, F_parts, gi_parts = [], [], []
    S = K[np.ix_(sep, sep)].copy() if nG else np.zeros((0, 0))
    q0 = woff = foff = gioff = 0
    for s, I in enumerate(interiors):
        m = I.size
        Kii = K[np.ix_(I, I)]
        Wi = np.linalg.inv(Kii)
        Wi = 0.5 * (Wi + Wi.T)
        Kig = K[np.ix_(I, sep)] if nG else np.zeros((m, 0))
        loc = np.flatnonzero(np.abs(Kig).sum(axis=0) != 0.0)
        Fi = Wi @ Kig[:, loc]
        if loc.size:
            S[np.ix_(loc, loc)] -= Kig[:, loc].T @ Fi
        meta[s] = (q0, m, woff, foff, loc.size, gioff)
        rowblk[q0:q0 + m] = s
        W_parts.append(np.asfortranarray(Wi).ravel(order="F"))
        F_parts.append(np.asfortranarray(Fi).ravel(order="F"))
        gi_parts.append(loc.astype(np.int32))
        q0 += m
        woff += m * m
        foff += m * loc.size
        gioff += loc.size
    if nG:
        Sinv = np.linalg.inv(S)
        Sinv = 0.5 * (Sinv + Sinv.T)
    else:
        Sinv = np.zeros((0, 0))
    # K[G, I] rows in permuted interior numbering (small CSR)
    gk_ptr = np.zeros(nG + 1, dtype=np.int32)
    gk_col, gk_val = [], []
    for g, v in enumerate(sep):
        cols = np.flatnonzero(K[v] != 0.0)
        cols = cols[~in_sep[cols]]
        gk_col.append(inv[cols].astype(np.int32))
        gk_val.append(K[v, cols])
        gk_ptr[g + 1] = gk_ptr[g] + cols.size
    cat = lambda parts, dt: (np.concatenate(parts).astype(dt) if parts and sum(p.size for p in parts) else np.zeros(0, dt))
    return dict(n=n, nI=nI, nG=nG, nparts=len(interiors), node=node, meta=meta.ravel(), rowblk=rowblk,
                W=cat(W_parts, np.float64), F=cat(F_parts, np.float64), gidx=cat(gi_parts, np.int32),
                Sinv=np.asfortranarray(Sinv).ravel(order="F"), gk_ptr=gk_ptr,
                gk_col=cat(gk_col, np.int32), gk_val=cat(gk_val, np.float64))


def solve_reference(pd: dict, b: np.ndarray) -> np.ndarray:
    """Numpy emulation of the kernel's solve phases (used by the CPU tests of the data layout)."""
    n, nI, nG = pd["n"], pd["nI"], pd["nG"]
    node = pd["node"]
    meta = pd["meta"].reshape(-1, 6)
    bp = b[node]
    y = np.zeros(nI)
    for q0, m, woff, foff, g, gioff in meta:
        W = pd["W"][woff:woff + m * m].reshape(m, m, order="F")
        y[q0:q0 + m] = W @ bp[q0:q0 + m]
    gv = bp[nI:].copy()
    for r in range(nG):
        s0, s1 = pd["gk_ptr"][r], pd["gk_ptr"][r + 1]
        gv[r] -= pd["gk_val"][s0:s1] @ y[pd["gk_col"][s0:s1]]
    xg = pd["Sinv"].reshape(nG, nG, order="F") @ gv if nG else gv
    xp = np.zeros(n)
    xp[nI:] = xg
    for q0, m, woff, foff, g, gioff in meta:
        F = pd["F"][foff:foff + m * g].reshape(m, g, order="F")
        xp[q0:q0 + m] = y[q0:q0 + m] - F @ xg[pd["gidx"][gioff:gioff + g]]
    x = np.zeros(n)
    x[node] = xp
    return x
