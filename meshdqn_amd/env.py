"""`Env2DAirfoil` - the reference's RL environment surface (Env2DAirfoil.py:42-602) on the MI355X kernels.

    env = Env2DAirfoil(config)            # config = parsed yaml (flow_config + agent_params)
    state = env.get_state()               # Data(x (N,2+3S) f32, edge_index (2,E) i64, edge_attr list)
    state, reward, done, info = env.step(action)   # action in [0, N_closest]; N_closest = "do nothing"

Same attribute names as the reference (`gt_drag, gt_lift, original_u, original_p, u, p, new_drags,
new_lifts, action_space.n, N_CLOSEST, coord_map, solver_steps, save_steps, flow_solver, ...`), same
error-code behaviour (0 ok / 1 already removed / 2 broken) and the same indexing quirks in the state
features.  Device work: IPCS ground-truth run (HIP evolve kernel), snapshot interpolation
(`mdq_interpolate_snapshots`), the 2S force integrals (`mdq_probe_forces`, one launch) and mesh
smoothing (`mdq_smooth_host`); vertex removal uses scipy's Qhull Delaunay exactly like the reference.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .data import Data
from .flow_solver import FlowSolver, Function, Mesh
from .mesh_ops import LightMeshBatch, SnapshotInterpolator, polygon_distance, remove_vertex_delaunay

device = torch.device("cpu")  # the reference forces the state tensors onto the CPU (Env2DAirfoil.py:39)


class Discrete:
    """gym.spaces.Discrete stand-in (only `.n` and `.sample()` are used by the reference)."""

    def __init__(self, n):
        self.n = int(n)

    def sample(self):
        return int(np.random.randint(self.n))


class Env2DAirfoil(object):
    """Environment to optimize mesh around a 2D airfoil."""

    def __init__(self, config, compute_device="cuda"):
        self.compute_device = torch.device(compute_device)
        self.flow_solver = FlowSolver(**config["flow_config"], device=self.compute_device)
        ap = config["agent_params"]
        self.coordinate_list = list(range(len(self.flow_solver.mesh.coordinates())))
        self.initial_num_node = len(self.coordinate_list)
        self.removable = np.argwhere(self.flow_solver.removable)[:, 0]
        self.mesh_map = {idx: int(rem) for idx, rem in enumerate(self.removable)}
        self.N_CLOSEST = ap["N_closest"]
        self.TIME_REWARD = ap["time_reward"]
        self.action_space = Discrete(self.N_CLOSEST)
        self.solver_steps = ap["solver_steps"]
        self.episodes = ap["episodes"]
        self.timesteps = ap["timesteps"]
        self.threshold = ap["threshold"]
        self.NEGATIVE_REWARD = -1.0
        self.removed_coordinates = []
        self.do_nothing_offset = 0
        self.gt_drag = np.array(ap["gt_drag"])
        self.gt_time = np.array(ap["gt_time"])
        self.gt_lift = np.array(ap.get("gt_lift", -1))
        self.u = ap["u"]
        self.p = ap["p"]
        self.original_u = ap["u"]
        self.original_p = ap["p"]
        self.save_steps = ap["save_steps"]
        self.goal_vertices = ap["goal_vertices"]
        self.plot_dir = ap.get("plot_dir", "")
        if not isinstance(self.u, int):
            self.u = [u.copy(deepcopy=True) for u in ap["u"]]
            self.p = [p.copy(deepcopy=True) for p in ap["p"]]
            self.original_u = [u.copy(deepcopy=True) for u in ap["u"]]
            self.original_p = [p.copy(deepcopy=True) for p in ap["p"]]
        self.POLYGON = False
        self.out_of_vertices = False
        self.reset()

    # ------------------------------------------------------------------
    def reset(self):
        if self.gt_drag.shape == ():
            self.gt_drag = np.array([self.gt_drag])
        if self.gt_time.shape == ():
            self.gt_time = np.array([self.gt_time])
        fs = self.flow_solver
        if (self.gt_drag[0] == -1) and (self.gt_time[0] == -1):
            # ground truth + snapshots: solver_steps IPCS steps, keep every save_steps-th (u, p, drag, lift)
            self.gt_drag, self.gt_lift, self.original_u, self.original_p, self.p, self.u = ([] for _ in range(6))
            done = 0
            while done < self.solver_steps:
                n = min(self.save_steps - (done % self.save_steps), self.solver_steps - done)
                u, p, drag, lift = fs.evolve(n)  # n steps in ONE kernel launch
                done += n
                if done % self.save_steps == 0:
                    self.gt_drag.append(drag)
                    self.gt_lift.append(lift)
                    self.original_u.append(u.copy(deepcopy=True))
                    self.original_p.append(p.copy(deepcopy=True))
                    self.u.append(u.copy(deepcopy=True))
                    self.p.append(p.copy(deepcopy=True))
            self.gt_drag = np.array(self.gt_drag)
            self.gt_lift = np.array(self.gt_lift)
            self.gt_time = np.array([fs.gtime])
        else:
            if isinstance(self.original_u, int):
                # the reference reloads snapshots/*.npy written by set_plot_dir (Env2DAirfoil.py:126-153)
                snap = os.path.join(self.plot_dir, "snapshots")
                save_us = np.load(os.path.join(snap, "save_velocities.npy"))
                save_ps = np.load(os.path.join(snap, "save_pressures.npy"))
                topo = fs.mesh.topology_
                self.original_u, self.original_p, self.u, self.p = [], [], [], []
                for i in range(int(np.ceil(self.solver_steps / self.save_steps))):
                    uf = Function(topo, torch.from_numpy(save_us[i].reshape(topo.np2, 2)).to(self.compute_device), "velocity")
                    pf = Function(topo, torch.from_numpy(save_ps[i].reshape(topo.nv)).to(self.compute_device), "pressure")
                    self.original_u.append(uf.copy(deepcopy=True))
                    self.u.append(uf.copy(deepcopy=True))
                    self.original_p.append(pf.copy(deepcopy=True))
                    self.p.append(pf.copy(deepcopy=True))
        topo = fs.mesh.topology_
        self._orig_topo = topo
        self._interp = SnapshotInterpolator(
            topo, topo.coords, torch.stack([u.data for u in self.original_u]),
            torch.stack([p.data for p in self.original_p]), device=self.compute_device)
        self._light = None
        self._calculate_velocities()
        self._calculate_pressures()
        self.steps = 0
        self.num_episodes = 0
        self.terminal = False
        self._get_distance_lookup()

    def return_vals(self):
        return self.gt_drag, self.gt_time

    def plot_state(self, title="{}", filename="initial_state"):
        """Env2DAirfoil.py:171-217: the mesh with removable / non-removable vertices, the N-closest selection and the
        edges of the state graph, written to `./{plot_dir}/{filename}.png` (matplotlib, when it is installed) or
        `.svg` (written directly otherwise).  Like the reference it recomputes the state first (`get_state`)."""
        state = self.get_state()
        mesh = self.flow_solver.mesh
        coords, cells = mesh.coordinates(), mesh.cells()
        removable = np.array(self.flow_solver.removable).astype(bool)
        sel = np.array(list(self.coord_map.values()), dtype=np.int64)
        mesh_edges = np.unique(np.sort(np.concatenate([cells[:, [0, 1]], cells[:, [0, 2]], cells[:, [1, 2]]]), axis=1), axis=0)
        ei = state.edge_index.cpu().numpy()
        graph_edges = np.stack([sel[ei[0]], sel[ei[1]]], axis=1) if ei.shape[1] else np.zeros((0, 2), np.int64)
        base = os.path.join(".", self.plot_dir, filename)
        os.makedirs(os.path.dirname(base) or ".", exist_ok=True)
        text = title.format(self.N_CLOSEST)
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            from matplotlib.collections import LineCollection
            fig, ax = plt.subplots(figsize=(10, 5))
            ax.add_collection(LineCollection(coords[mesh_edges], colors="#888888", linewidths=0.75, zorder=0))
            ax.scatter(coords[:, 0], coords[:, 1], color=np.array(["r", "k"])[removable.astype(int)], s=6, zorder=1)
            ax.add_collection(LineCollection(coords[graph_edges], colors="b", linewidths=0.75, zorder=2))
            ax.scatter(coords[sel, 0], coords[sel, 1], color="b", s=6, zorder=3)
            ax.set_title(text, fontsize=18, y=0.975)
            ax.autoscale()
            ax.set_axis_off()
            fig.savefig(base + ".png", bbox_inches="tight")
            plt.close(fig)
            return base + ".png"
        except ImportError:
            pass
        lo, hi = coords.min(axis=0), coords.max(axis=0)
        sc = 1000.0 / (hi[0] - lo[0])
        px = lambda q: ((q[0] - lo[0]) * sc, (hi[1] - q[1]) * sc + 30.0)   # noqa: E731
        out = [f'<svg xmlns="http://www.w3.org/2000/svg" width="1000" height="{(hi[1] - lo[1]) * sc + 40:.0f}">',
               f'<text x="500" y="20" text-anchor="middle" font-size="18">{text}</text>']
        for edges, col in ((mesh_edges, "#888888"), (graph_edges, "blue")):
            for a, b in edges:
                (x1, y1), (x2, y2) = px(coords[a]), px(coords[b])
                out.append(f'<line x1="{x1:.1f}" y1="{y1:.1f}" x2="{x2:.1f}" y2="{y2:.1f}" stroke="{col}" stroke-width="0.75"/>')
        in_state = np.zeros(len(coords), bool)
        in_state[sel] = True
        for i, q in enumerate(coords):
            x1, y1 = px(q)
            col = "blue" if in_state[i] else ("black" if removable[i] else "red")
            out.append(f'<circle cx="{x1:.1f}" cy="{y1:.1f}" r="1.5" fill="{col}"/>')
        out.append("</svg>")
        with open(base + ".svg", "w") as f:
            f.write("\n".join(out))
        return base + ".svg"

    # ------------------------------------------------------------------
    def _get_distance_lookup(self):
        coords = self.flow_solver.mesh.coordinates()
        if not self.POLYGON:
            not_removable = np.argwhere(~np.array(self.flow_solver.removable, dtype=bool))[:, 0]
            bc = coords[not_removable]
            sel = (bc[:, 0] > -0.5) & (bc[:, 0] < 3) & (bc[:, 1] > -0.5) & (bc[:, 1] < 0.5)
            self.polygon = bc[sel].copy()
            self.POLYGON = True
        self.distance_lookup = list(polygon_distance(self.polygon, coords[self.removable]))

    def _n_closest(self):
        self.coordinate_list = list(range(len(self.flow_solver.mesh.coordinates())))
        self.removable = np.argwhere(self.flow_solver.removable)[:, 0]
        self.mesh_map = dict(zip(range(len(self.removable)), self.removable.tolist()))
        self._get_distance_lookup()
        dist_idxs = np.argsort(self.distance_lookup)
        self.n_closest = dist_idxs[self.do_nothing_offset:self.N_CLOSEST + self.do_nothing_offset]
        if len(self.n_closest) < self.N_CLOSEST:
            print("OUT OF VERTICES")
            self.out_of_vertices = True
        mapping = self.removable[self.n_closest]
        self.coord_map = dict(zip(range(len(self.n_closest)), mapping.tolist()))
        self.inv_coord_map = dict(zip(mapping.tolist(), range(len(self.n_closest))))

    def get_state(self):
        edge_index, edge_attr = [], []
        self._n_closest()
        vals = np.array(list(self.coord_map.values())).astype(int)
        mesh_cells = self.flow_solver.mesh.cells()
        good = np.argwhere(np.all(np.isin(mesh_cells, vals), axis=1))[:, 0]
        X = self.flow_solver.mesh.coordinates()
        if good.size:
            gc = mesh_cells[good]
            inv = np.full(X.shape[0], -1, dtype=np.int64)
            inv[vals] = np.arange(vals.size)
            ids = inv[gc]  # (m,3)
            edge_index = np.stack([ids[:, [0, 0, 1]].ravel(), ids[:, [1, 2, 2]].ravel()])
            c = X[gc]
            l = np.stack([np.linalg.norm(c[:, 0] - c[:, 1], axis=1), np.linalg.norm(c[:, 0] - c[:, 2], axis=1),
                          np.linalg.norm(c[:, 1] - c[:, 2], axis=1)], axis=1)
            edge_attr = l.ravel().tolist()
            edge_index = torch.from_numpy(edge_index).long()
        else:
            edge_index = torch.zeros((2, 0), dtype=torch.long)
        S = self.velocities.shape[0]
        n = len(self.n_closest)
        x = torch.zeros((self.N_CLOSEST, 3 * S + 2), dtype=torch.float)
        # NB (kept from the reference, Env2DAirfoil.py:285-288): rows are indexed by n_closest (the rank
        # inside the removable list), not by the vertex id, and the velocity block is a raw reshape
        x[:n, :2] = torch.from_numpy(X[self.n_closest])
        x[:n, 2:2 * S + 2] = torch.from_numpy(self.velocities[:, self.n_closest, :].reshape(n, -1))
        x[:n, 2 * S + 2:] = torch.from_numpy(self.pressures[:, self.n_closest][:, :, 0].T)
        return Data(x=x, edge_index=edge_index, edge_attr=edge_attr).to(device)

    # ------------------------------------------------------------------
    def step(self, action):
        broken = False
        if action == self.action_space.n:  # no removal: shift the N-closest window
            self.do_nothing_offset += 1
            removed = 0
        else:
            removed = self._remove_vertex(action)
        state = self.get_state()
        if self.out_of_vertices:
            print("OUT OF VERTICES")
            removed = 2
        if removed == 0:
            rew, broken, self.terminal = self.calculate_reward()
            if self.terminal:
                self.rew = 0.5 * self.NEGATIVE_REWARD
            if broken:
                rew = self.NEGATIVE_REWARD
                self.terminal = True
        elif removed == 1:
            rew = self.NEGATIVE_REWARD
        else:
            rew = self.NEGATIVE_REWARD
            self.terminal = True
            broken = True
        self.steps += 1
        if self.steps >= self.timesteps:
            self.terminal = True
            self.episodes += 1
        if isinstance(rew, float) and np.isnan(rew):
            rew = self.NEGATIVE_REWARD
        return state, rew, self.terminal, {}

    def _mesh_batch(self):
        if self._light is None:
            topo = self.flow_solver.mesh.topology_
            self._light = LightMeshBatch([topo], [topo.coords], self.flow_solver.mu, device=self.compute_device)
        return self._light

    def calculate_reward(self):
        try:
            lb = self._mesh_batch()
            topo = self.flow_solver.mesh.topology_
            S = len(self.u)
            ub = torch.zeros((1, S, lb.N2, 2), dtype=torch.float64, device=self.compute_device)
            pb = torch.zeros((1, S, lb.cap["NV"]), dtype=torch.float64, device=self.compute_device)
            for i, (u, p) in enumerate(zip(self.u, self.p)):
                ub[0, i, :topo.np2] = u.data
                pb[0, i, :topo.nv] = p.data
            drag, lift = lb.probe_forces(ub, pb)  # 2S surface integrals in one launch
            self.new_drags = drag[0].cpu().numpy()
            self.new_lifts = lift[0].cpu().numpy()
        except Exception:
            print("\n\nSAMPLING BROKE\n\n")
            return self.NEGATIVE_REWARD, True, True
        drag_factor = -2 * np.log(0.5) / self.threshold
        error_val = np.linalg.norm(np.abs(self.gt_drag - self.new_drags) / np.abs(self.gt_drag))
        drag_reward = 2 * np.exp(-drag_factor * error_val) - 1
        time_reward = (self.initial_num_node - len(self.coordinate_list)) * self.TIME_REWARD
        acc_thresh = any(np.abs(np.abs(self.gt_drag - self.new_drags) / self.gt_drag) > self.threshold)
        vert_thresh = len(self.flow_solver.mesh.coordinates()) < self.goal_vertices * self.initial_num_node
        if vert_thresh:
            print("\nMAXIMUM REMOVALS REACHED\n")
        return float(drag_reward + time_reward), False, bool(acc_thresh or vert_thresh)

    def set_plot_dir(self, plot_dir):
        self.plot_dir = plot_dir
        os.makedirs(os.path.join(plot_dir, "snapshots"), exist_ok=True)
        np.save(os.path.join(plot_dir, "snapshots", "velocities.npy"), self.velocities)
        np.save(os.path.join(plot_dir, "snapshots", "pressures.npy"), self.pressures)
        np.save(os.path.join(plot_dir, "snapshots", "save_velocities.npy"),
                np.array([u.vector().get_local() for u in self.original_u]))
        np.save(os.path.join(plot_dir, "snapshots", "save_pressures.npy"),
                np.array([p.vector().get_local() for p in self.original_p]))

    # ------------------------------------------------------------------
    def _remove_vertex(self, selected_coord=None):
        try:
            selected_coord = self.coord_map[selected_coord]
        except KeyError:
            print("RAN OUT OF VERTICES")
            return 2
        idx = self.coordinate_list.index(selected_coord)
        topo = self.flow_solver.mesh.topology_
        boundary_vertices = np.flatnonzero(topo.on_boundary)
        coords = topo.coords
        self.removed_coordinates.append(coords[idx].copy())
        del self.coordinate_list[idx]
        try:
            new_coords, cells = remove_vertex_delaunay(coords, boundary_vertices, idx)
        except ValueError:  # Qhull could not triangulate
            self.coordinate_list.insert(selected_coord, selected_coord)
            print("\nMESH BROKE, COULDN'T TRIANGULATE")
            return 2
        try:
            mesh = Mesh(new_coords, cells)
        except ValueError:
            self.coordinate_list.insert(selected_coord, selected_coord)
            return 2
        return self._check_mesh(mesh, selected_coord)

    def _calculate_velocities(self):
        self.velocities = np.array([u.vertex_values() for u in self.u])  # (S, nv, 2)

    def _calculate_pressures(self):
        self.pressures = np.array([p.vertex_values() for p in self.p])[:, :, np.newaxis]  # (S, nv, 1)

    def _check_mesh(self, mesh, selected_coord):
        if selected_coord in self.removable:
            old = self.flow_solver.snapshot()       # (the reference keeps `old_mesh`, Env2DAirfoil.py:552)
            self.flow_solver.remesh(mesh)  # smooth(50) again, new removable / probes
            topo = self.flow_solver.mesh.topology_
            try:
                out_u, out_p = self._interp.interpolate([topo], [topo.coords])
            except Exception:
                print("INTERPOLATION BROKE")
                # back to the old mesh (Env2DAirfoil.py:556-558) - together with everything remesh() derived from the
                # new one, so that u / p / velocities and the solver's mesh keep belonging together
                self.flow_solver.restore(old)
                self.coordinate_list.insert(selected_coord, selected_coord)
                return 2
            for i in range(len(self.original_u)):
                self.u[i] = Function(topo, out_u[0, i, :topo.np2].clone(), "velocity")
                self.p[i] = Function(topo, out_p[0, i, :topo.nv].clone(), "pressure")
            self._light = None
            self._calculate_velocities()
            self._calculate_pressures()
            self.removable = np.argwhere(self.flow_solver.removable)[:, 0]
            return 0
        else:
            self.coordinate_list.insert(selected_coord, selected_coord)
            print("\nMESH BROKE. SKIPPING VERTEX REMOVAL\n")
            return 2
