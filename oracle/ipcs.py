"""Oracle: the reference's `FlowSolver` (IPCS, sparse direct solves) on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows `flow_solver.py:49-191` (construction: load, smooth(50), tag, assemble
three matrices once, LU-factorise once because the yaml key `solver_type` is
ignored and `la_solve` defaults to 'lu') and `flow_solver.py:362-396`
(`evolve`: three RHS assemblies + three back-substitutions, then drag / lift).
"""
import numpy as np
import scipy.sparse.linalg as spla

from .fem import TaylorHood
from .mesh import OracleMesh


class OracleFlowSolver:
    def __init__(self, coords, cells, mu=1e-3, rho=1.0, dt=1e-3, smooth=True, smooth_iters=50,
                 factorize=True, inflow=None):
        # inflow: None = the reference's constant parabola; callable profile(x, y, t) -> x-velocity = a time dependent
        # inflow expression (flow_solver.py:70-73; `inflow.time = gtime` at the top of evolve, :369-371)
        self.inflow = inflow
        self.mesh = OracleMesh(coords, cells)
        if smooth:
            self.mesh.smooth(smooth_iters)
        self.mu, self.rho, self.dt = mu, rho, dt
        self.removable = self.mesh.removable()
        self.th = TaylorHood(self.mesh, mu=mu, rho=rho, dt=dt)
        th = self.th
        self.A1, self.lift1 = th.apply_bc_symmetric(th.A1_full, th.bcu_dofs, th.bcu_vals)
        self.A2, self.lift2 = th.apply_bc_symmetric(th.K1, th.bcp_dofs, th.bcp_vals)
        self.A3, self.lift3 = th.apply_bc_symmetric(th.Mv, th.bcu_dofs, th.bcu_vals)
        if factorize:
            self.lu1 = spla.splu(self.A1)
            self.lu2 = spla.splu(self.A2)
            self.lu3 = spla.splu(self.A3)
        self.u_n = np.zeros(2 * th.np2)
        self.p_n = np.zeros(th.nv)
        self.gtime = 0.0
        self.accumulated_drag, self.accumulated_lift = [], []

    # individual right-hand sides (also used by the per-kernel parity tests)
    def rhs1(self, u_n, p_n):
        th = self.th
        b = th.R1 @ u_n - self.rho * th.convection(u_n) + th.Dv @ p_n - self.lift1
        b[th.bcu_dofs] = th.bcu_vals
        return b

    def rhs2(self, u_s, p_n):
        th = self.th
        b = th.K1 @ p_n - (1.0 / self.dt) * (th.Dv.T @ u_s) - self.lift2
        b[th.bcp_dofs] = th.bcp_vals
        return b

    def rhs3(self, u_s, p_new, p_n):
        th = self.th
        b = th.Mv @ u_s - self.dt * (th.Gv @ (p_new - p_n)) - self.lift3
        b[th.bcu_dofs] = th.bcu_vals
        return b

    def evolve(self):
        self.gtime += self.dt
        if self.inflow is not None:
            th = self.th
            xy = th.dof_coords[th.inlet_dofs]
            th.bcu_vals[th.inlet_pos] = self.inflow(xy[:, 0], xy[:, 1], self.gtime)
            g = np.zeros(th.A1_full.shape[0])
            g[th.bcu_dofs] = th.bcu_vals
            self.lift1 = th.A1_full @ g       # the lifting of the symmetric elimination follows the boundary values
            self.lift3 = th.Mv @ g
        u_s = self.lu1.solve(self.rhs1(self.u_n, self.p_n))
        p_new = self.lu2.solve(self.rhs2(u_s, self.p_n))
        u_new = self.lu3.solve(self.rhs3(u_s, p_new, self.p_n))
        self.u_n, self.p_n = u_new, p_new
        drag, lift = self.th.forces(u_new, p_new)
        self.accumulated_drag.append(drag)
        self.accumulated_lift.append(lift)
        return u_new, p_new, drag, lift
