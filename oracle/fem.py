"""Oracle: Taylor-Hood (P2/P1) discretisation of the reference's IPCS forms.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows `flow_solver.py:85-144` (spaces, forms F1/a2/L2/a3/L3, Dirichlet BCs,
SystemAssembler symmetric elimination), `flow_solver.py:33-44` (inflow
parabola) and `probes.py:23-50` (drag / lift integrals).  UFL conventions
(SURVEY Appendix A.2): nabla_grad(u)[i,j] = d_i u_j, epsilon = sym(nabla_grad),
(nabla_grad(U)*n)_i = sum_j d_i U_j n_j.

Dof layout (ours; DOLFIN's internal numbering is not reproducible):
  scalar P2: [0,nv) vertex dofs, nv+e edge-midpoint dofs; velocity = [ux | uy].
  local P2 basis: phi_k = l_k(2 l_k - 1), phi_{3+k} = 4 l_a l_b on the edge
  opposite local vertex k.
"""
import numpy as np
import scipy.sparse as sp

from .mesh import OracleMesh

# ---------------------------------------------------------------- quadrature


def _gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def tri_quadrature(n=4):
    """Collapsed (Duffy) Gauss rule on the unit triangle; exact to degree 2n-2."""
    gx, gw = _gauss01(n)
    pts, wts = [], []
    for i in range(n):
        for j in range(n):
            x = gx[i]
            y = gx[j] * (1.0 - x)
            pts.append((x, y))
            wts.append(gw[i] * gw[j] * (1.0 - x))
    return np.array(pts), np.array(wts)


_EDGE_AB = ((1, 2), (0, 2), (0, 1))


def p1_basis(xi, eta):
    lam = np.array([1.0 - xi - eta, xi, eta])
    dlam = np.array([[-1.0, -1.0], [1.0, 0.0], [0.0, 1.0]])
    return lam, dlam


def p2_basis(xi, eta):
    """Values (6,) and reference gradients (6,2) of the P2 basis at a point."""
    lam, dlam = p1_basis(xi, eta)
    phi = np.zeros(6)
    dphi = np.zeros((6, 2))
    for k in range(3):
        phi[k] = lam[k] * (2.0 * lam[k] - 1.0)
        dphi[k] = (4.0 * lam[k] - 1.0) * dlam[k]
    for k, (a, b) in enumerate(_EDGE_AB):
        phi[3 + k] = 4.0 * lam[a] * lam[b]
        dphi[3 + k] = 4.0 * (lam[a] * dlam[b] + lam[b] * dlam[a])
    return phi, dphi


# ---------------------------------------------------------------- discretisation


class TaylorHood:
    """All IPCS operators on one mesh, assembled with numpy/scipy."""

    def __init__(self, mesh: OracleMesh, mu=1e-3, rho=1.0, dt=1e-3):
        self.mesh = mesh
        self.mu, self.rho, self.dt = float(mu), float(rho), float(dt)
        m = mesh
        self.nv, self.ne, self.nt = m.nv, m.ne, m.nt
        self.np2 = m.nv + m.ne
        self.cell_dofs = np.concatenate([m.cells, m.nv + m.cell_edges], axis=1)  # (nt,6)
        X = m.coords[m.cells]  # (nt,3,2)
        # affine map x = x0 + J [xi,eta]
        J = np.stack([X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]], axis=2)  # (nt,2,2) columns
        det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
        Jinv = np.empty_like(J)
        Jinv[:, 0, 0] = J[:, 1, 1] / det
        Jinv[:, 0, 1] = -J[:, 0, 1] / det
        Jinv[:, 1, 0] = -J[:, 1, 0] / det
        Jinv[:, 1, 1] = J[:, 0, 0] / det
        self.J, self.Jinv, self.absdet = J, Jinv, np.abs(det)
        self.dof_coords = np.concatenate([m.coords, 0.5 * (m.coords[m.edges[:, 0]] + m.coords[m.edges[:, 1]])])
        self._assemble()
        self._boundary_conditions()

    # physical gradient: grad phi = Jinv^T dphi_ref  ->  (nt,6,2)
    def _phys_grad(self, dref):
        return np.einsum("tca,ic->tia", self.Jinv, dref)

    def _assemble(self):
        nt = self.nt
        pts, wts = tri_quadrature(4)
        M = np.zeros((nt, 6, 6))
        Kab = np.zeros((nt, 2, 2, 6, 6))  # Kab[a,b][i,j] = int d_a phi_i d_b phi_j
        Dc = np.zeros((nt, 2, 6, 3))  # int psi_j d_c phi_i
        Gc = np.zeros((nt, 2, 6, 3))  # int phi_i d_c psi_j
        K1 = np.zeros((nt, 3, 3))
        self._quad = []
        for (xi, eta), w in zip(pts, wts):
            phi, dphi = p2_basis(xi, eta)
            psi, dpsi = p1_basis(xi, eta)
            g2 = self._phys_grad(dphi)  # (nt,6,2)
            g1 = self._phys_grad(dpsi)  # (nt,3,2)
            ww = w * self.absdet
            M += ww[:, None, None] * np.outer(phi, phi)[None]
            Kab += np.einsum("t,tia,tjb->tabij", ww, g2, g2)
            Dc += np.einsum("t,tic,j->tcij", ww, g2, psi)
            Gc += np.einsum("t,i,tjc->tcij", ww, phi, g1)
            K1 += np.einsum("t,tia,tja->tij", ww, g1, g1)
            self._quad.append((phi, g2, ww))
        self.Me, self.Kabe, self.Dce, self.Gce, self.K1e = M, Kab, Dc, Gc, K1
        cd = self.cell_dofs
        cv = self.mesh.cells
        n2, n1 = self.np2, self.nv

        def asm(E, rows, cols, shape):
            r = np.repeat(rows, cols.shape[1], axis=1).ravel()
            c = np.tile(cols, (1, rows.shape[1])).ravel()
            return sp.coo_matrix((E.ravel(), (r, c)), shape=shape).tocsr()

        self.M = asm(M, cd, cd, (n2, n2))
        self.K = [[asm(Kab[:, a, b], cd, cd, (n2, n2)) for b in range(2)] for a in range(2)]
        self.D = [asm(Dc[:, c], cd, cv, (n2, n1)) for c in range(2)]
        self.G = [asm(Gc[:, c], cd, cv, (n2, n1)) for c in range(2)]
        self.K1 = asm(K1, cv, cv, (n1, n1))
        L = self.K[0][0] + self.K[1][1]
        # eps:eps block (test comp c, trial comp d) = 1/2 (delta_cd L + K^{dc})
        Keps = sp.bmat([[0.5 * (L + self.K[0][0]), 0.5 * self.K[1][0]],
                        [0.5 * self.K[0][1], 0.5 * (L + self.K[1][1])]]).tocsr()
        B = self._outflow_term()
        Mv = sp.block_diag([self.M, self.M]).tocsr()
        self.Mv = Mv
        self.Avisc = (self.mu * Keps - 0.5 * self.mu * B).tocsr()
        self.A1_full = (self.rho / self.dt) * Mv + self.Avisc
        self.R1 = ((self.rho / self.dt) * Mv - self.Avisc).tocsr()
        self.Dv = sp.vstack(self.D).tocsr()  # (2 n2, n1): pressure -> momentum
        self.Gv = sp.vstack(self.G).tocsr()

    def _outflow_term(self):
        """B[(c,i),(d,j)] = int_{outflow facets} phi_i (d_c phi_j) n_d ds
        (`- dot(mu*nabla_grad(U)*n, v)*ds`, `flow_solver.py:109`; only rows of
        unconstrained boundary dofs matter and those all sit on tag-3 facets)."""
        m = self.mesh
        n2 = self.np2
        tags = m.facet_tags()
        gx, gw = _gauss01(3)
        rows, cols, vals = [], [], []
        for e, t in tags.items():
            if t != 3:
                continue
            (c, k) = m.edge_cells[e][0]
            normal, length, refpts = self._facet_geometry(c, k)
            dofs = self.cell_dofs[c]
            E = np.zeros((2, 2, 6, 6))
            for s, w in zip(gx, gw):
                xi, eta = refpts(s)
                phi, dphi = p2_basis(xi, eta)
                g = dphi @ self.Jinv[c]  # (6,2): row i = grad phi_i (physical)
                for cc in range(2):
                    for dd in range(2):
                        E[cc, dd] += w * length * np.outer(phi, g[:, cc]) * normal[dd]
            for cc in range(2):
                for dd in range(2):
                    for i in range(6):
                        for j in range(6):
                            if E[cc, dd, i, j] != 0.0:
                                rows.append(cc * n2 + dofs[i])
                                cols.append(dd * n2 + dofs[j])
                                vals.append(E[cc, dd, i, j])
        return sp.coo_matrix((vals, (rows, cols)), shape=(2 * n2, 2 * n2)).tocsr()

    def _facet_geometry(self, c, k):
        """Outward unit normal, length and reference parametrisation of the
        local edge k (opposite local vertex k) of cell c."""
        m = self.mesh
        a_l, b_l = _EDGE_AB[k]
        ref = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
        xa, xb = m.coords[m.cells[c, a_l]], m.coords[m.cells[c, b_l]]
        xo = m.coords[m.cells[c, k]]
        t = xb - xa
        length = np.sqrt(t[0] * t[0] + t[1] * t[1])
        n = np.array([t[1], -t[0]]) / length
        if np.dot(n, xo - xa) > 0:
            n = -n
        ra, rb = ref[a_l], ref[b_l]

        def refpts(s):
            p = ra + s * (rb - ra)
            return p[0], p[1]

        return n, length, refpts

    # ------------------------------------------------------------------
    def inflow_profile(self, xy):
        """`constant_profile` (`flow_solver.py:33-44`)."""
        bot = self.mesh.coords[:, 1].min()
        top = self.mesh.coords[:, 1].max()
        H = top - bot
        Um = 1.5
        return -4.0 * Um * (xy[:, 1] - bot) * (xy[:, 1] - top) / H / H

    def _boundary_conditions(self):
        m = self.mesh
        nv, n2 = self.nv, self.np2
        tags = m.facet_tags()
        self.tags = tags
        # velocity: bcu = [inlet(tag 2), airfoil, walls] - later wins
        val, src = {}, {}
        for want, fn in ((2, "inlet"), (1, "zero"), (0, "zero")):
            for e, t in tags.items():
                if t != want:
                    continue
                for d in (m.edges[e, 0], m.edges[e, 1], nv + e):
                    if fn == "inlet":
                        val[int(d)] = float(self.inflow_profile(self.dof_coords[d:d + 1])[0])
                    else:
                        val[int(d)] = 0.0
                    src[int(d)] = fn
        sd = np.array(sorted(val), dtype=np.int64)
        self.bcu_scalar_dofs = sd
        # positions (in bcu_vals) and dof ids of the dofs that carry the inflow profile (flow_solver.py:369-371)
        self.inlet_pos = np.array([i for i, d in enumerate(sd) if src[int(d)] == "inlet"], dtype=np.int64)
        self.inlet_dofs = sd[self.inlet_pos]
        self.bcu_dofs = np.concatenate([sd, n2 + sd])
        gx = np.array([val[int(d)] for d in sd])
        # inlet gives ux = parabola, uy = 0 ; airfoil / walls give zero for both
        self.bcu_vals = np.concatenate([gx, np.zeros_like(gx)])
        # pressure: p = 0 on outflow facets (vertices only)
        pd = set()
        for e, t in tags.items():
            if t == 3:
                pd.add(int(m.edges[e, 0]))
                pd.add(int(m.edges[e, 1]))
        self.bcp_dofs = np.array(sorted(pd), dtype=np.int64)
        self.bcp_vals = np.zeros(len(self.bcp_dofs))

    @staticmethod
    def apply_bc_symmetric(A, dofs, vals):
        """Global form of SystemAssembler's symmetric elimination (unit diagonal).
        Returns (A_bc, lift) with b_bc = b - lift; b_bc[dofs] = vals."""
        n = A.shape[0]
        g = np.zeros(n)
        g[dofs] = vals
        lift = A @ g
        keep = np.ones(n)
        keep[dofs] = 0.0
        Dk = sp.diags(keep)
        A_bc = (Dk @ A @ Dk + sp.diags(1.0 - keep)).tocsc()
        return A_bc, lift

    # ------------------------------------------------------------------
    def convection(self, u):
        """N(u)^c_i = int (u . grad u_c) phi_i  (`rho*dot(dot(u_n, nabla_grad(u_n)), v)*dx`)."""
        n2 = self.np2
        cd = self.cell_dofs
        ux, uy = u[:n2][cd], u[n2:][cd]  # (nt,6)
        out = np.zeros(2 * n2)
        ex = np.zeros((self.nt, 6))
        ey = np.zeros((self.nt, 6))
        for phi, g2, ww in self._quad:
            vx = ux @ phi
            vy = uy @ phi
            dux = np.einsum("ti,tia->ta", ux, g2)  # grad ux (nt,2)
            duy = np.einsum("ti,tia->ta", uy, g2)
            cx = vx * dux[:, 0] + vy * dux[:, 1]
            cy = vx * duy[:, 0] + vy * duy[:, 1]
            ex += (ww * cx)[:, None] * phi[None]
            ey += (ww * cy)[:, None] * phi[None]
        np.add.at(out, cd.ravel(), ex.ravel())
        np.add.at(out, n2 + cd.ravel(), ey.ravel())
        return out

    # ------------------------------------------------------------------
    def airfoil_facets(self):
        return [e for e, t in self.tags.items() if t == 1]

    def forces(self, u, p, facet_tag=1):
        """(drag, lift) = int_{tag} ((2 mu sym(grad u) - p I) n) ds  (`probes.py:23-50`)."""
        m = self.mesh
        n2 = self.np2
        gx, gw = _gauss01(2)
        drag = lift = 0.0
        for e, t in self.tags.items():
            if t != facet_tag:
                continue
            (c, k) = m.edge_cells[e][0]
            n, length, refpts = self._facet_geometry(c, k)
            dofs = self.cell_dofs[c]
            ux, uy = u[dofs], u[n2 + dofs]
            pv = p[m.cells[c]]
            for s, w in zip(gx, gw):
                xi, eta = refpts(s)
                phi, dphi = p2_basis(xi, eta)
                psi, _ = p1_basis(xi, eta)
                g = dphi @ self.Jinv[c]
                gux = ux @ g  # (d_x ux, d_y ux)
                guy = uy @ g
                pp = pv @ psi
                sxx = 2 * self.mu * gux[0] - pp
                sxy = self.mu * (gux[1] + guy[0])
                syy = 2 * self.mu * guy[1] - pp
                drag += w * length * (sxx * n[0] + sxy * n[1])
                lift += w * length * (sxy * n[0] + syy * n[1])
        return drag, lift
