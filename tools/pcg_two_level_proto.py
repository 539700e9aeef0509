"""numpy prototype: Jacobi-CG against the two-level additive preconditioner (Jacobi + piecewise-constant coarse space over
geometric aggregates, exact coarse solve) on the pressure system of the ys930 family - the lab mesh, its red refinement
(BASELINE configs[4], 3 322 vertices) and the second refinement (12 924 vertices) - for the right-hand side of an IPCS step
in developing flow (a few oracle steps from rest) warm-started from the previous pressure.  Iterations at the stopping test
of the kernels (preconditioned... here: residual 2-norm of the Jacobi-scaled system, rtol 1e-10), and a cost model:
an iteration of the device CG is one operator application + 2-3 reductions; the two-level one adds a restriction, an
(n_agg x n_agg) coarse product and a prolongation.  VERDICT r4 #4: "measure the two-level preconditioner on the refined
meshes, where Jacobi-CG needs 320 / 833 iterations".   python tools/pcg_two_level_proto.py [levels]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import scipy.sparse as sp
from make_refined_fixtures import red_refine
from oracle.ipcs import OracleFlowSolver
from oracle.mesh import OracleMesh


def pcg(A, b, x0, apply_prec, rtol, maxit=5000):
    x = x0.copy()
    r = b - A @ x
    z = apply_prec(r)
    p = z.copy()
    rz = r @ z
    bb = np.sqrt(b @ b)
    it = 0
    while np.sqrt(r @ r) > rtol * bb and it < maxit:
        q = A @ p
        al = rz / (p @ q)
        x += al * p
        r -= al * q
        z = apply_prec(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
        it += 1
    return x, it


def aggregates(xy, nagx, nagy):
    """NAGX strips of equal population by x, every strip cut into NAGY cells of equal population by y (the device kernel's rule)."""
    n = len(xy)
    ox = np.argsort(xy[:, 0], kind="stable")
    strip = np.empty(n, np.int64)
    strip[ox] = np.arange(n) * nagx // n
    agg = np.empty(n, np.int64)
    for s in range(nagx):
        idx = np.flatnonzero(strip == s)
        oy = idx[np.argsort(xy[idx, 1], kind="stable")]
        agg[oy] = s * nagy + np.arange(len(oy)) * nagy // max(len(oy), 1)
    return agg


def main():
    levels = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    z = np.load(os.path.join(R, "tests", "golden", "ys930.npz"))
    base = OracleMesh(z["coords"], z["cells"]); base.smooth(50)
    coords, cells = base.coords, z["cells"]
    for lev in range(levels + 1):
        if lev:
            coords, cells = red_refine(coords, cells)
        t0 = time.time()
        fs = OracleFlowSolver(coords, cells, smooth=(lev > 0))
        for _ in range(3):
            u_s = fs.lu1.solve(fs.rhs1(fs.u_n, fs.p_n)); p_prev = fs.p_n.copy()
            fs.evolve()
        # the pressure system of the NEXT step, warm start = current pressure
        u_s = fs.lu1.solve(fs.rhs1(fs.u_n, fs.p_n))
        b = fs.rhs2(u_s, fs.p_n)
        A = fs.A2.tocsr()
        d = np.sqrt(A.diagonal())
        As = sp.diags(1 / d) @ A @ sp.diags(1 / d)       # Jacobi-scaled (unit diagonal): what the kernels iterate on
        bs, x0 = b / d, fs.p_n * d
        n = A.shape[0]
        xy = fs.mesh.coords
        row = f"level {lev}: {n} vertices (setup {time.time() - t0:.0f}s):"
        _, it0 = pcg(As, bs, x0, lambda r: r, 1e-10)
        row += f" Jacobi-CG {it0} iterations;"
        for nagx, nagy in ((8, 7), (16, 14), (32, 28), (64, 56)):
            if nagx * nagy * 4 > n:
                continue
            agg = aggregates(xy, nagx, nagy)
            P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nagx * nagy))
            Ac = (P.T @ As @ P).toarray()
            Aci = np.linalg.inv(Ac)
            _, it = pcg(As, bs, x0, lambda r: r + P @ (Aci @ (P.T @ r)), 1e-10)
            row += f" two-level {nagx}x{nagy}={nagx * nagy}: {it} ({it0 / it:.2f}x fewer);"
        print(row, flush=True)


if __name__ == "__main__":
    main()
