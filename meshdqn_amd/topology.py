"""Host-side mesh topology for the HIP path (vectorised numpy; product code).

What DOLFIN derives implicitly for the reference (`flow_solver.py:59-86,
123-132,194-226`) and what our HIP kernels need as index arrays:

  * ordered cells, unique edges, boundary edges / vertices   (Mesh.order, BoundaryMesh)
  * exterior-facet tags walls 0 / airfoil 1 / inflow 2 / outflow 3 / other 4
  * `removable` with the numpy-`in` quirk of `flow_solver.py:77-78`
  * Taylor-Hood dof maps (scalar P2 = vertices + edge midpoints, P1 = vertices)
  * CSR sparsity patterns of the P2 and P1 operators, and the deterministic
    gather maps  nnz <- (cell,i,j)  and  dof <- (cell,i)  used by the assembly
    and right-hand-side kernels (no atomics: fixed summation order)
  * Dirichlet dof flags / values (bcu = [inlet, airfoil, walls], bcp = [outflow])

This module never imports `oracle/`.
"""
from __future__ import annotations

import numpy as np

DOLFIN_EPS = 3.0e-16

TAG_WALL, TAG_AIRFOIL, TAG_INFLOW, TAG_OUTFLOW, TAG_OTHER = 0, 1, 2, 3, 4

# local edge k is opposite local vertex k
_EDGE_A = np.array([1, 0, 0])
_EDGE_B = np.array([2, 2, 1])


class MeshTopology:
    """Topology + index maps of one triangle mesh (coordinates may still move:
    smoothing never changes connectivity)."""

    def __init__(self, coords: np.ndarray, cells: np.ndarray):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.sort(np.asarray(cells, dtype=np.int64), axis=1)
        if coords.ndim != 2 or coords.shape[1] != 2:
            raise ValueError("coords must be (nv,2)")
        if cells.ndim != 2 or cells.shape[1] != 3:
            raise ValueError("cells must be (nt,3)")
        if cells.size and (cells.min() < 0 or cells.max() >= coords.shape[0]):
            raise ValueError("cell vertex index out of range")
        self.coords = coords
        self.cells = cells
        self.nv = coords.shape[0]
        self.nt = cells.shape[0]
        self._edges()

    # ------------------------------------------------------------------
    def _edges(self):
        nv, c = self.nv, self.cells
        a = c[:, _EDGE_A]  # (nt,3) smaller endpoint (cells are sorted)
        b = c[:, _EDGE_B]
        keys = (a * nv + b).ravel()
        uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
        order = np.argsort(first, kind="stable")  # number edges by first appearance
        rank = np.empty_like(order)
        rank[order] = np.arange(order.size)
        self.cell_edges = rank[inv].reshape(self.nt, 3)
        uk = uniq[order]
        self.edges = np.stack([uk // nv, uk % nv], axis=1)
        self.ne = self.edges.shape[0]
        cnt = np.bincount(self.cell_edges.ravel(), minlength=self.ne)
        if cnt.size and cnt.max() > 2:
            raise ValueError("non-manifold mesh: an edge is shared by more than two cells")
        self.edge_ncells = cnt
        self.boundary_edges = np.flatnonzero(cnt == 1)
        on_b = np.zeros(nv, dtype=bool)
        on_b[self.edges[self.boundary_edges].ravel()] = True
        self.on_boundary = on_b
        # owner (cell, local edge) of every edge: first occurrence
        flat_first = first[order]
        self.edge_cell = flat_first // 3
        self.edge_local = flat_first % 3
        # dof maps
        self.np2 = nv + self.ne
        self.cell_dofs = np.concatenate([c, nv + self.cell_edges], axis=1)

    def permuted(self, perm: np.ndarray) -> "MeshTopology":
        """Same mesh, same vertex / edge / dof numbering, cells stored in the order `perm` (new cell i = old cell
        perm[i]).  Only the cell-indexed arrays change; vectors over dofs are unaffected."""
        import copy
        perm = np.asarray(perm, dtype=np.int64)
        if perm.shape != (self.nt,) or not np.array_equal(np.sort(perm), np.arange(self.nt)):
            raise ValueError("perm must be a permutation of the cells")
        t = copy.copy(self)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(self.nt)
        t.cells = self.cells[perm]
        t.cell_edges = self.cell_edges[perm]
        t.cell_dofs = self.cell_dofs[perm]
        t.edge_cell = inv[self.edge_cell]
        return t

    # ------------------------------------------------------------------
    def vertex_adjacency(self):
        """CSR vertex->neighbour vertices (edge order) and vertex->(cell,local)."""
        nv = self.nv
        e = self.edges
        src = np.concatenate([e[:, 0], e[:, 1]])
        dst = np.concatenate([e[:, 1], e[:, 0]])
        eid = np.concatenate([np.arange(self.ne), np.arange(self.ne)])
        o = np.lexsort((eid, src))
        nbr_ptr = np.zeros(nv + 1, dtype=np.int64)
        np.cumsum(np.bincount(src, minlength=nv), out=nbr_ptr[1:])
        nbr = dst[o]
        cv = self.cells.ravel()
        cid = np.repeat(np.arange(self.nt), 3)
        o2 = np.lexsort((cid, cv))
        vc_ptr = np.zeros(nv + 1, dtype=np.int64)
        np.cumsum(np.bincount(cv, minlength=nv), out=vc_ptr[1:])
        vc = (cid * 3 + np.tile(np.arange(3), self.nt))[o2]
        return nbr_ptr, nbr, vc_ptr, vc

    # ------------------------------------------------------------------
    def removable(self, coords=None):
        """`coord not in bmesh.coordinates()` (`flow_solver.py:75-78`): numpy
        `__contains__` => not ((bcoords == coord).any())."""
        x = self.coords if coords is None else coords
        bc = x[self.on_boundary]
        hit_x = np.isin(x[:, 0], bc[:, 0])
        hit_y = np.isin(x[:, 1], bc[:, 1])
        return ~(hit_x | hit_y)

    # ------------------------------------------------------------------
    def facet_tags(self, coords=None):
        """Tag of every boundary edge (aligned with `self.boundary_edges`)."""
        x = self.coords if coords is None else coords
        e = self.edges[self.boundary_edges]
        pa, pb = x[e[:, 0]], x[e[:, 1]]
        pts = np.stack([pa, pb, 0.5 * (pa + pb)], axis=1)  # (nb,3,2)
        X, Y = pts[..., 0], pts[..., 1]
        E = DOLFIN_EPS
        walls = ((Y > 0.5 - 2 * E) | (Y < -0.5 + 2 * E)).all(axis=1)
        airfoil = ((X < 3.0 - E) & (X > -0.5 + E) & (Y < 0.5 - E) & (Y > -0.5 + E)).all(axis=1)
        inflow = (X < -0.5 + E).all(axis=1)
        outflow = (X > 3.0 - 2 * E).all(axis=1)
        tags = np.full(e.shape[0], TAG_OTHER, dtype=np.int32)
        tags[walls] = TAG_WALL
        tags[airfoil] = TAG_AIRFOIL
        tags[inflow] = TAG_INFLOW
        tags[outflow] = TAG_OUTFLOW
        return tags

    # ------------------------------------------------------------------
    def dof_coords(self, coords=None):
        x = self.coords if coords is None else coords
        return np.concatenate([x, 0.5 * (x[self.edges[:, 0]] + x[self.edges[:, 1]])])

    def boundary_conditions(self, coords=None, inflow=None, time=0.0):
        """Dirichlet data of `flow_solver.py:123-132`.

        Returns dict with
          bcu_flag (np2,) uint8, bcu_gx (np2,) f8 (x-velocity value; y value is 0
          for every BC of the reference), bcp_flag (nv,) uint8, inlet_dofs (scalar P2 dofs whose value is the inflow
          profile's: the inlet dofs the airfoil / wall conditions do not override).
        `inflow`: None = the reference's time independent parabola (`constant_profile`, flow_solver.py:33-44), or a
        callable profile(x, y, t) -> x-velocity at the dof coordinates (the reference's time dependent expression,
        flow_solver.py:70-73,369-371).
        """
        x = self.coords if coords is None else coords
        tags = self.facet_tags(x)
        be = self.boundary_edges
        nv = self.nv
        dofx = self.dof_coords(x)
        flag = np.zeros(self.np2, dtype=np.uint8)
        gx = np.zeros(self.np2)
        bot, top = x[:, 1].min(), x[:, 1].max()
        H = top - bot
        Um = 1.5
        # list order [inlet, airfoil, walls]: later wins on shared dofs
        inlet = np.zeros(self.np2, dtype=bool)
        for want in (TAG_INFLOW, TAG_AIRFOIL, TAG_WALL):
            es = be[tags == want]
            d = np.concatenate([self.edges[es, 0], self.edges[es, 1], nv + es])
            flag[d] = 1
            if want == TAG_INFLOW:
                y = dofx[d, 1]
                if inflow is None:
                    gx[d] = -4.0 * Um * (y - bot) * (y - top) / H / H
                else:
                    gx[d] = np.asarray(inflow(dofx[d, 0], y, float(time)), dtype=np.float64)
                inlet[d] = True
            else:
                gx[d] = 0.0
                inlet[d] = False
        pflag = np.zeros(nv, dtype=np.uint8)
        es = be[tags == TAG_OUTFLOW]
        pflag[self.edges[es].ravel()] = 1
        return dict(bcu_flag=flag, bcu_gx=gx, bcp_flag=pflag, tags=tags, inlet_dofs=np.flatnonzero(inlet))

    # ------------------------------------------------------------------
    def facets(self, tags, want):
        """(cell, local_edge) of all boundary edges with tag `want`, in edge-id order."""
        es = self.boundary_edges[tags == want]
        return np.stack([self.edge_cell[es], self.edge_local[es]], axis=1).astype(np.int32), es

    # ------------------------------------------------------------------
    @staticmethod
    def _pattern(rows_local, cols_local, nrows, ncols):
        """CSR pattern + gather map for element matrices with local shape (nr,nc).

        rows_local (nt,nr), cols_local (nt,nc).  Returns rowptr, colidx,
        asm_ptr (nnz+1), asm_src (nt*nr*nc) where asm_src lists, per non-zero,
        the flat element-matrix slots  e*nr*nc + i*nc + j  that sum into it
        (ascending e: deterministic order)."""
        nt, nr = rows_local.shape
        nc = cols_local.shape[1]
        r = np.repeat(rows_local, nc, axis=1).ravel()
        c = np.tile(cols_local, (1, nr)).ravel()
        keys = r * ncols + c
        uniq, inv = np.unique(keys, return_inverse=True)
        rows = uniq // ncols
        colidx = (uniq % ncols).astype(np.int32)
        rowptr = np.zeros(nrows + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows, minlength=nrows), out=rowptr[1:])
        order = np.argsort(inv, kind="stable")
        asm_src = order.astype(np.int32)
        asm_ptr = np.zeros(uniq.size + 1, dtype=np.int32)
        np.cumsum(np.bincount(inv, minlength=uniq.size), out=asm_ptr[1:])
        return rowptr, colidx, asm_ptr, asm_src

    def patterns(self):
        cd, cv = self.cell_dofs, self.cells
        out = {}
        out["p2"] = self._pattern(cd, cd, self.np2, self.np2)
        out["p1"] = self._pattern(cv, cv, self.nv, self.nv)
        return out

    @staticmethod
    def sell_layout(rowptr, colidx, C=64):
        """SELL-C layout (slices of C consecutive rows, column-major inside a
        slice, slice width = longest row of the slice; no row permutation).

        Returns sl_off (nslices+1, offsets in entries), sl_col (padded column
        indices; padding points at the row itself) and pos (nnz,) = position of
        every CSR non-zero inside the SELL arrays."""
        n = rowptr.size - 1
        lens = np.diff(rowptr).astype(np.int64)
        ns = (n + C - 1) // C
        padded = np.zeros(ns * C, dtype=np.int64)
        padded[:n] = lens
        width = padded.reshape(ns, C).max(axis=1)
        sl_off = np.zeros(ns + 1, dtype=np.int64)
        np.cumsum(width * C, out=sl_off[1:])
        rows = np.repeat(np.arange(n), lens)
        j = np.arange(rowptr[-1]) - np.repeat(rowptr[:-1].astype(np.int64), lens)
        pos = sl_off[rows // C] + j * C + rows % C
        total = int(sl_off[-1])
        # padding entries reference their own row (always a valid index) with value 0
        allrows = np.minimum(np.arange(ns * C), max(n - 1, 0))
        sl_col = np.empty(total, dtype=np.int32)
        for s_ in range(ns):
            w = int(width[s_])
            sl_col[sl_off[s_]:sl_off[s_ + 1]] = np.tile(allrows[s_ * C:(s_ + 1) * C], w)
        sl_col[pos] = colidx[:rowptr[-1]]
        return sl_off.astype(np.int32), sl_col, pos.astype(np.int32)

    @staticmethod
    def _dof_gather(dofs_local, ndofs):
        """dof <- list of flat slots e*nl + i (ascending e)."""
        flat = dofs_local.ravel()
        order = np.argsort(flat, kind="stable").astype(np.int32)
        ptr = np.zeros(ndofs + 1, dtype=np.int32)
        np.cumsum(np.bincount(flat, minlength=ndofs), out=ptr[1:])
        return ptr, order

    def matfree_maps(self, chunk=1024):
        """Tile maps of the matrix-free operator application (one chunk = `chunk`
        consecutive triangles = one LDS tile of 6*chunk element results):
          scat (6, nt): position of result i of triangle e inside its chunk's tile;
                        tile entries are ordered by (destination row, triangle) so that
          tptr (nchunks, np2+1): row r owns tile entries [tptr[c,r], tptr[c,r+1]) of chunk c,
                        in ascending triangle order (deterministic summation)."""
        nt, n2 = self.nt, self.np2
        nch = max((nt + chunk - 1) // chunk, 1)
        rows = self.cell_dofs  # (nt,6)
        e_idx = np.repeat(np.arange(nt)[:, None], 6, axis=1)
        scat = np.zeros((nt, 6), dtype=np.int32)
        tptr = np.zeros((nch, n2 + 1), dtype=np.int32)
        for c in range(nch):
            lo, hi = c * chunk, min((c + 1) * chunk, nt)
            r = rows[lo:hi].ravel()
            e = e_idx[lo:hi].ravel()
            order = np.lexsort((e, r))
            pos = np.empty(order.size, dtype=np.int32)
            pos[order] = np.arange(order.size, dtype=np.int32)
            scat[lo:hi] = pos.reshape(hi - lo, 6)
            np.cumsum(np.bincount(r, minlength=n2), out=tptr[c, 1:])
        return np.ascontiguousarray(scat.T), tptr

    def matfree_packed(self, cell_outflow, chunk=1024):
        """(6, nt) int32 words  dof | tile_pos << 12 | (outflow_edge + 1) << 28 (word 0 only):
        everything a thread needs to apply one triangle's operator, in 6 coalesced loads."""
        if self.np2 > 4096 or 6 * chunk > 65536:
            raise ValueError("packed matrix-free metadata needs np2 <= 4096")
        scat, tptr = self.matfree_maps(chunk)
        w = self.cell_dofs.T.astype(np.int64) | (scat.astype(np.int64) << 12)
        w[0] |= (np.asarray(cell_outflow, dtype=np.int64) + 1) << 28
        return w.astype(np.int32), tptr

    def dof_gathers(self):
        return dict(p2=self._dof_gather(self.cell_dofs, self.np2),
                    p1=self._dof_gather(self.cells, self.nv))


def morton_cell_order(coords: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """Permutation of the cells along a Morton curve of their centroids (16 bits per axis over the bounding box of the
    vertices, ties in ascending cell id) - the host twin of `mdq_flow_sort_cells` (csrc/mdq_tilemaps.hip): chunks of 1 024
    consecutive triangles then share their rows, what the tile maps of the element-tile operator modes 5 / 7 need (red-refined
    ys930: 2 200 touched rows per chunk and 1.06 chunks per row, against 5 000 and 2.4 in the refinement's own order or in the
    conflict-free order of the LDS-atomic mode - more than the kernels' LDS stage of a chunk's input rows holds).  Returns
    `perm` with new_cells = cells[perm]."""
    coords = np.asarray(coords, dtype=np.float64)
    lo, hi = coords.min(0), coords.max(0)
    tri = coords[np.asarray(cells)]
    cen = ((tri[:, 0] + tri[:, 1]) + tri[:, 2]) * (1.0 / 3.0)
    scale = np.where(hi > lo, 65536.0 / np.where(hi > lo, hi - lo, 1.0), 0.0)
    q = np.clip(((cen - lo) * scale).astype(np.int64), 0, 65535)

    def part(x):
        x = x.astype(np.uint64) & np.uint64(0xFFFF)
        for sh, mk in ((8, 0x00FF00FF), (4, 0x0F0F0F0F), (2, 0x33333333), (1, 0x55555555)):
            x = (x | (x << np.uint64(sh))) & np.uint64(mk)
        return x
    key = part(q[:, 0]) | (part(q[:, 1]) << np.uint64(1))
    return np.argsort(key, kind="stable").astype(np.int64)


def conflict_free_cell_order(cells: np.ndarray, edges_of_cells: np.ndarray | None = None, block: int = 64) -> np.ndarray:
    """Permutation of the cells such that the cells of one block of `block` consecutive positions share no vertex
    (hence no P2 dof): a wave of the matrix-free kernels (lane = cell position % 64 inside a round) then never issues
    two LDS atomics to the same address in one instruction.  Greedy first-fit over blocks; cells that fit nowhere
    (rare) fill the remaining holes.  Returns `perm` with new_cells = cells[perm]."""
    nt = cells.shape[0]
    nblocks = (nt + block - 1) // block
    used = [set() for _ in range(nblocks)]
    fill = [[] for _ in range(nblocks)]
    cap = [min(block, nt - b * block) for b in range(nblocks)]
    left = []
    start = 0
    for t in range(nt):
        vs = cells[t]
        placed = False
        for k in range(nblocks):
            b = (start + k) % nblocks
            if len(fill[b]) < cap[b] and not (vs[0] in used[b] or vs[1] in used[b] or vs[2] in used[b]):
                fill[b].append(t)
                used[b].update((int(vs[0]), int(vs[1]), int(vs[2])))
                placed = True
                break
        start = (start + 1) % nblocks
        if not placed:
            left.append(t)
    for t in left:
        for b in range(nblocks):
            if len(fill[b]) < cap[b]:
                fill[b].append(t)
                break
    return np.array([t for f in fill for t in f], dtype=np.int64)
