"""Oracle: mesh topology, DOLFIN-style smoothing and boundary tagging.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Deliberately written
with plain Python loops and dictionaries so that it is an independent check
of the vectorised host code in `meshdqn_amd/` and of the HIP kernels.

Follows:
  * `flow_solver.py:59-67`   mesh load + `mesh.smooth(50)`
  * `flow_solver.py:75-78`   `removable` quirk (`coord not in ndarray`)
  * `flow_solver.py:9-30,194-226`  SubDomain tagging of exterior facets
  * DOLFIN `MeshSmoothing::smooth`, `Mesh.order()`, `BoundaryMesh`
"""
import numpy as np

DOLFIN_EPS = 3.0e-16


class OracleMesh:
    def __init__(self, coords, cells):
        self.coords = np.array(coords, dtype=np.float64)
        # DOLFIN orders cell vertex lists ascending (mesh.order()).
        self.cells = np.sort(np.array(cells, dtype=np.int64), axis=1)
        self.nv = self.coords.shape[0]
        self.nt = self.cells.shape[0]
        self._build()

    def _build(self):
        edge_id = {}
        edges = []
        edge_cells = []
        cell_edges = np.zeros((self.nt, 3), dtype=np.int64)
        # local edge k is opposite local vertex k
        for c in range(self.nt):
            v = self.cells[c]
            for k, (a, b) in enumerate(((1, 2), (0, 2), (0, 1))):
                key = (int(v[a]), int(v[b]))
                if key not in edge_id:
                    edge_id[key] = len(edges)
                    edges.append(key)
                    edge_cells.append([])
                e = edge_id[key]
                edge_cells[e].append((c, k))
                cell_edges[c, k] = e
        self.edges = np.array(edges, dtype=np.int64)
        self.ne = len(edges)
        self.edge_cells = edge_cells
        self.cell_edges = cell_edges
        self.boundary_edges = [e for e in range(self.ne) if len(edge_cells[e]) == 1]
        on_b = np.zeros(self.nv, dtype=bool)
        for e in self.boundary_edges:
            on_b[self.edges[e, 0]] = True
            on_b[self.edges[e, 1]] = True
        self.on_boundary = on_b
        # vertex -> neighbours (edge order) and vertex -> cells
        nbrs = [[] for _ in range(self.nv)]
        for a, b in edges:
            nbrs[a].append(b)
            nbrs[b].append(a)
        self.nbrs = nbrs
        vcells = [[] for _ in range(self.nv)]
        for c in range(self.nt):
            for k in range(3):
                vcells[self.cells[c, k]].append((c, k))
        self.vcells = vcells

    # ------------------------------------------------------------------
    def smooth(self, num_iterations=50):
        """DOLFIN `MeshSmoothing::smooth` (Gauss-Seidel, index order)."""
        x = self.coords
        for _ in range(num_iterations):
            for v in range(self.nv):
                if self.on_boundary[v]:
                    continue
                p = x[v].copy()
                xx = np.zeros(2)
                for n in self.nbrs[v]:
                    xx += x[n]
                xx /= float(len(self.nbrs[v]))
                rmin = 0.0
                for (c, k) in self.vcells[v]:
                    o = [self.cells[c, j] for j in range(3) if j != k]
                    a, b = x[o[0]], x[o[1]]
                    t = b - a
                    nrm = np.array([t[1], -t[0]])
                    nrm /= np.sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1])
                    r = abs(nrm[0] * (p[0] - a[0]) + nrm[1] * (p[1] - a[1]))
                    rmin = r if rmin == 0.0 else min(rmin, r)
                d = xx - p
                r = np.sqrt(d[0] * d[0] + d[1] * d[1])
                if r < DOLFIN_EPS:
                    continue
                step = min(0.5 * rmin, r)
                x[v] = p + step * d / r
        return self

    # ------------------------------------------------------------------
    def removable(self):
        """`coord not in bmesh.coordinates()` - numpy `__contains__` quirk
        (`flow_solver.py:75-78`): true-if-any-scalar-matches."""
        bc = self.coords[self.on_boundary]
        out = np.zeros(self.nv, dtype=bool)
        for v in range(self.nv):
            out[v] = not bool((bc == self.coords[v]).any())
        return out

    # ------------------------------------------------------------------
    def facet_tags(self):
        """Tag per boundary edge: 0 walls, 1 airfoil, 2 inflow, 3 outflow, 4 other
        (`flow_solver.py:194-226`; later marks override earlier ones)."""
        EPS = DOLFIN_EPS

        def walls(p):
            return (p[1] > 0.5 - 2 * EPS) or (p[1] < -0.5 + 2 * EPS)

        def airfoil(p):
            return (p[0] < 3.0 - EPS) and (p[0] > -0.5 + EPS) and (p[1] < 0.5 - EPS) and (p[1] > -0.5 + EPS)

        def inflow(p):
            return p[0] < -0.5 + EPS

        def outflow(p):
            return p[0] > 3.0 - 2 * EPS

        tags = {}
        for e in self.boundary_edges:
            a = self.coords[self.edges[e, 0]]
            b = self.coords[self.edges[e, 1]]
            pts = (a, b, 0.5 * (a + b))
            t = 4
            for val, fn in ((0, walls), (1, airfoil), (2, inflow), (3, outflow)):
                if all(fn(p) for p in pts):
                    t = val
            tags[e] = t
        return tags
