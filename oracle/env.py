"""Oracle: the reference's `Env2DAirfoil` (vertex removal -> Delaunay -> smoothing -> snapshot
interpolation -> reward -> N-closest graph state) restated with numpy / scipy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference has no test,
fixture or golden vector for this part and dolfin / shapely / torch_geometric cannot be imported
here; the third-party behaviour is restated from its documentation at the cited call sites
(SURVEY.md Appendix A.1, A.3, A.4).

Follows Env2DAirfoil.py:47-164 (__init__/reset), :220-315 (distance lookup, get_state, _n_closest),
:318-428 (step, calculate_reward), :452-522 (_remove_vertex, vertex values), :547-602 (_check_mesh).
"""
import numpy as np
from scipy.spatial import Delaunay

from .fem import TaylorHood, p1_basis, p2_basis
from .ipcs import OracleFlowSolver
from .mesh import OracleMesh


def point_segment_distance(p, a, b):
    ab = b - a
    t = np.dot(p - a, ab) / np.dot(ab, ab)
    t = min(1.0, max(0.0, t))
    q = a + t * ab
    return float(np.hypot(*(p - q)))


def point_in_polygon(p, poly):
    """Even-odd rule (what GEOS' point-in-ring test computes for simple rings)."""
    x, y = p
    inside = False
    n = len(poly)
    for i in range(n):
        x1, y1 = poly[i]
        x2, y2 = poly[(i + 1) % n]
        if (y1 > y) != (y2 > y):
            xin = x1 + (y - y1) * (x2 - x1) / (y2 - y1)
            if x < xin:
                inside = not inside
    return inside


def polygon_distance(poly, p):
    """shapely `Polygon(poly).distance(Point(p))`: 0 inside/on the ring, else distance to the ring."""
    if point_in_polygon(p, poly):
        return 0.0
    n = len(poly)
    return min(point_segment_distance(p, poly[i], poly[(i + 1) % n]) for i in range(n))


class P2P1Evaluator:
    """`Function.interpolate(other)` / `u(x, allow_extrapolation=True)` on a fixed source mesh:
    locate a containing cell (closest cell if none), evaluate the P2 / P1 polynomial."""

    def __init__(self, th: TaylorHood):
        self.th = th
        m = th.mesh
        self.X0 = m.coords[m.cells[:, 0]]
        self.Jinv = th.Jinv

    def locate(self, pts):
        """Return (cell, xi, eta) per point."""
        cells = np.empty(len(pts), dtype=np.int64)
        ref = np.empty((len(pts), 2))
        for k, p in enumerate(pts):
            d = p[None, :] - self.X0
            # reference coords: [xi,eta] = J^-1 (x - x0)
            xi = self.Jinv[:, 0, 0] * d[:, 0] + self.Jinv[:, 0, 1] * d[:, 1]
            eta = self.Jinv[:, 1, 0] * d[:, 0] + self.Jinv[:, 1, 1] * d[:, 1]
            lam = np.stack([1.0 - xi - eta, xi, eta], axis=1)
            viol = np.minimum(lam.min(axis=1), 0.0)  # 0 inside, negative outside
            c = int(np.argmax(viol))  # first cell with the smallest violation (0 if some cell contains p)
            cells[k] = c
            ref[k] = (xi[c], eta[c])
        return cells, ref

    def eval_p2(self, u, cells, ref):
        """u: velocity dof vector [ux|uy] (2*np2). Returns (npts,2)."""
        th = self.th
        n2 = th.np2
        out = np.empty((len(cells), 2))
        for k, (c, (xi, eta)) in enumerate(zip(cells, ref)):
            phi, _ = p2_basis(xi, eta)
            dofs = th.cell_dofs[c]
            out[k, 0] = u[dofs] @ phi
            out[k, 1] = u[n2 + dofs] @ phi
        return out

    def eval_p1(self, p, cells, ref):
        th = self.th
        out = np.empty(len(cells))
        for k, (c, (xi, eta)) in enumerate(zip(cells, ref)):
            psi, _ = p1_basis(xi, eta)
            out[k] = p[th.mesh.cells[c]] @ psi
        return out


class OracleEnv:
    NEGATIVE_REWARD = -1.0

    def __init__(self, coords, cells, agent_params, flow_params=None, solver_params=None, snapshots=None):
        fp = dict(mu=1e-3, rho=1.0)
        fp.update(flow_params or {})
        sp = dict(dt=1e-3, smooth=True)
        sp.update(solver_params or {})
        self.fp, self.sp = fp, sp
        self.flow = OracleFlowSolver(coords, cells, mu=fp["mu"], rho=fp["rho"], dt=sp["dt"], smooth=sp["smooth"],
                                     factorize=snapshots is None)
        self.smooth = sp["smooth"]
        ap = agent_params
        self.N_CLOSEST = ap["N_closest"]
        self.TIME_REWARD = ap["time_reward"]
        self.solver_steps, self.save_steps = ap["solver_steps"], ap["save_steps"]
        self.timesteps, self.threshold = ap["timesteps"], ap["threshold"]
        self.goal_vertices = ap["goal_vertices"]
        self.coordinate_list = list(range(self.flow.mesh.nv))
        self.initial_num_node = len(self.coordinate_list)
        self.removable = np.argwhere(self.flow.removable)[:, 0]
        self.do_nothing_offset = 0
        self.out_of_vertices = False
        self.removed_coordinates = []
        # ---- reset(): ground truth + snapshots on the ORIGINAL mesh (Env2DAirfoil.py:102-153)
        self.orig_th = self.flow.th
        self.evaluator = P2P1Evaluator(self.orig_th)
        if snapshots is None:
            self.gt_drag, self.gt_lift, self.original_u, self.original_p = [], [], [], []
            for i in range(self.solver_steps):
                u, p, drag, lift = self.flow.evolve()
                if (i + 1) % self.save_steps == 0:
                    self.gt_drag.append(drag)
                    self.gt_lift.append(lift)
                    self.original_u.append(u.copy())
                    self.original_p.append(p.copy())
        else:
            self.gt_drag, self.gt_lift = list(snapshots["gt_drag"]), list(snapshots["gt_lift"])
            self.original_u = [np.array(a) for a in snapshots["u"]]
            self.original_p = [np.array(a) for a in snapshots["p"]]
        self.gt_drag = np.array(self.gt_drag)
        self.gt_lift = np.array(self.gt_lift)
        self.u = [a.copy() for a in self.original_u]
        self.p = [a.copy() for a in self.original_p]
        self.cur_th = self.orig_th
        self._vertex_values()
        self.steps = 0
        self.terminal = False
        self._polygon = None
        self._get_distance_lookup()

    # ------------------------------------------------------------------
    def _vertex_values(self):
        """`_calculate_velocities/_pressures` (:515-522): values of the current functions at the
        current vertices = vertex dofs."""
        nv, n2 = self.cur_th.nv, self.cur_th.np2
        self.velocities = np.array([np.stack([u[:nv], u[n2:n2 + nv]], axis=1) for u in self.u])  # (S,nv,2)
        self.pressures = np.array([p[:nv] for p in self.p])[:, :, None]  # (S,nv,1)

    def _get_distance_lookup(self):
        mesh = self.flow.mesh
        coords = mesh.coords
        rem = np.array(self.flow.removable)
        if self._polygon is None:
            not_removable = np.argwhere(~rem)[:, 0]
            bc = coords[not_removable]
            sel = (bc[:, 0] > -0.5) & (bc[:, 0] < 3) & (bc[:, 1] > -0.5) & (bc[:, 1] < 0.5)
            self._polygon = bc[sel].copy()
        self.distance_lookup = [polygon_distance(self._polygon, c) for c in coords[self.removable]]

    def _n_closest(self):
        self.coordinate_list = list(range(self.flow.mesh.nv))
        self.removable = np.argwhere(self.flow.removable)[:, 0]
        self._get_distance_lookup()
        dist_idxs = np.argsort(self.distance_lookup)
        self.n_closest = dist_idxs[self.do_nothing_offset:self.N_CLOSEST + self.do_nothing_offset]
        if len(self.n_closest) < self.N_CLOSEST:
            self.out_of_vertices = True
        mapping = self.removable[self.n_closest]
        self.coord_map = dict(zip(range(len(self.n_closest)), mapping.tolist()))
        self.inv_coord_map = dict(zip(mapping.tolist(), range(len(self.n_closest))))

    def get_state(self):
        """Returns dict(x (N,2+3S) f32, edge_index (2,E) i64, edge_attr list) - Env2DAirfoil.py:244-290
        including its two indexing quirks (features indexed by n_closest, raw reshape of velocities)."""
        self._n_closest()
        mesh = self.flow.mesh
        vals = np.array(list(self.coord_map.values()), dtype=np.int64)
        cells = mesh.cells
        good = np.argwhere(np.all(np.isin(cells, vals), axis=1))[:, 0]
        edge_index, edge_attr = [], []
        X = mesh.coords
        for idx in good:
            c = cells[idx]
            i1, i2, i3 = (self.inv_coord_map[int(v)] for v in c)
            c1, c2, c3 = X[c[0]], X[c[1]], X[c[2]]
            edge_attr += [np.linalg.norm(c1 - c2), np.linalg.norm(c1 - c3), np.linalg.norm(c2 - c3)]
            edge_index += [[i1, i2], [i1, i3], [i2, i3]]
        S = self.velocities.shape[0]
        x = np.zeros((self.N_CLOSEST, 3 * S + 2), dtype=np.float32)
        nc = self.n_closest
        x[:, :2] = X[nc]
        x[:, 2:2 * S + 2] = self.velocities[:, nc, :].reshape(self.N_CLOSEST, -1)
        x[:, 2 * S + 2:] = self.pressures[:, nc][:, :, 0].T
        ei = np.array(edge_index, dtype=np.int64).T if edge_index else np.zeros((2, 0), dtype=np.int64)
        return dict(x=x, edge_index=ei, edge_attr=edge_attr)

    # ------------------------------------------------------------------
    def calculate_reward(self):
        nd, nl = [], []
        for u, p in zip(self.u, self.p):
            d, l = self.cur_th.forces(u, p)
            nd.append(d)
            nl.append(l)
        self.new_drags, self.new_lifts = np.array(nd), np.array(nl)
        drag_factor = -2 * np.log(0.5) / self.threshold
        error_val = np.linalg.norm(np.abs(self.gt_drag - self.new_drags) / np.abs(self.gt_drag))
        drag_reward = 2 * np.exp(-drag_factor * error_val) - 1
        time_reward = (self.initial_num_node - len(self.coordinate_list)) * self.TIME_REWARD
        acc_thresh = any(np.abs(np.abs(self.gt_drag - self.new_drags) / self.gt_drag) > self.threshold)
        vert_thresh = self.flow.mesh.nv < self.goal_vertices * self.initial_num_node
        return drag_reward + time_reward, False, bool(acc_thresh or vert_thresh)

    def _remove_vertex(self, action):
        try:
            selected = self.coord_map[action]
        except KeyError:
            return 2
        idx = self.coordinate_list.index(selected)
        mesh = self.flow.mesh
        boundary_vertices = np.flatnonzero(mesh.on_boundary).copy()
        coords = mesh.coords
        self.removed_coordinates.append(coords[idx].copy())
        boundary_vertices[boundary_vertices > idx] -= 1
        keep = [i for i in range(len(coords)) if i != idx]
        del self.coordinate_list[idx]
        coords = coords[keep]
        try:
            tri = Delaunay(coords)
        except ValueError:
            self.coordinate_list.insert(selected, selected)
            return 2
        cells = tri.simplices
        cells = cells[np.sum(np.isin(cells, boundary_vertices), axis=1) != 3]
        return self._check_mesh(coords, cells, selected)

    def _check_mesh(self, coords, cells, selected):
        if selected not in self.removable:
            self.coordinate_list.insert(selected, selected)
            return 2
        # flow_solver.remesh (non-deploy): new mesh, smooth(50), new spaces / probes
        m = OracleMesh(coords, cells)
        if self.smooth:
            m.smooth(50)
        self.flow.mesh = m
        self.flow.removable = m.removable()
        th = TaylorHood(m, mu=self.fp["mu"], rho=self.fp["rho"], dt=self.sp["dt"])
        self.cur_th = th
        # interpolate every snapshot FROM THE ORIGINAL MESH onto the new P2 / P1 dofs
        pts2 = th.dof_coords
        c2, r2 = self.evaluator.locate(pts2)
        nv = th.nv
        for i in range(len(self.original_u)):
            uv = self.evaluator.eval_p2(self.original_u[i], c2, r2)
            self.u[i] = np.concatenate([uv[:, 0], uv[:, 1]])
            self.p[i] = self.evaluator.eval_p1(self.original_p[i], c2[:nv], r2[:nv])
        self._vertex_values()
        self.removable = np.argwhere(self.flow.removable)[:, 0]
        return 0

    def step(self, action):
        broken = False
        if action == self.N_CLOSEST:
            self.do_nothing_offset += 1
            removed = 0
        else:
            removed = self._remove_vertex(action)
        state = self.get_state()
        if self.out_of_vertices:
            removed = 2
        if removed == 0:
            rew, broken, self.terminal = self.calculate_reward()
            if broken:
                rew = self.NEGATIVE_REWARD
                self.terminal = True
        elif removed == 1:
            rew = self.NEGATIVE_REWARD
        else:
            rew = self.NEGATIVE_REWARD
            self.terminal = True
        self.steps += 1
        if self.steps >= self.timesteps:
            self.terminal = True
        return state, rew, self.terminal, {}
