"""Mesh-side device operations of the environment step (host wrappers over the C ABI).

  * `LightMeshBatch`   - just enough of a mesh batch on the device for the force probes
                         (`mdq_probe_forces`) on freshly coarsened meshes.
  * `SnapshotInterpolator` - `Function.interpolate` of the stored snapshots from the ORIGINAL mesh
                         onto coarsened meshes (`mdq_interpolate_snapshots`): uniform location grid
                         built once on the host, point location + P2/P1 evaluation on the GPU.
  * `remove_vertex_delaunay` - vertex removal + global re-triangulation exactly as the reference does it
                         (`scipy.spatial.Delaunay` on the host, Env2DAirfoil.py:480-496).
  * `polygon_distance`  - vectorised `shapely.Polygon.distance(Point)` for the N-closest ranking.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch
from scipy.spatial import Delaunay

from . import _lib
from .topology import TAG_AIRFOIL, MeshTopology


class LightMeshBatch:
    def __init__(self, topos: Sequence[MeshTopology], coords: Sequence[np.ndarray], mu: float, device="cuda",
                 capacities: dict | None = None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.topos = list(topos)
        B = len(self.topos)
        afs = []
        for t, x in zip(self.topos, coords):
            tags = t.facet_tags(x)
            af, _ = t.facets(tags, TAG_AIRFOIL)
            afs.append(af)
        cap = dict(NV=max(t.nv for t in topos), NT=max(t.nt for t in topos), NE=max(t.ne for t in topos),
                   NAF=max(max(a.shape[0] for a in afs), 1))
        if capacities:
            for k, v in capacities.items():
                cap[k] = max(cap[k], int(v))
        self.cap, self.B = cap, B
        NV, NT, NE, NAF = cap["NV"], cap["NT"], cap["NE"], cap["NAF"]
        self.N2 = NV + NE
        h_coords = np.zeros((B, NV, 2))
        h_cd = np.zeros((B, 6, NT), np.int32)
        h_af = np.zeros((B, NAF, 2), np.int32)
        for b, (t, x, af) in enumerate(zip(self.topos, coords, afs)):
            h_coords[b, :t.nv] = x
            h_cd[b, :, :t.nt] = t.cell_dofs.T
            h_af[b, :af.shape[0]] = af
        dev = self.device
        self.t = dict(coords=torch.from_numpy(h_coords).to(dev), cell_dofs=torch.from_numpy(h_cd).to(dev),
                      af_facets=torch.from_numpy(h_af).to(dev),
                      nv=torch.tensor([t.nv for t in topos], dtype=torch.int32, device=dev),
                      nt=torch.tensor([t.nt for t in topos], dtype=torch.int32, device=dev),
                      ne=torch.tensor([t.ne for t in topos], dtype=torch.int32, device=dev),
                      naf=torch.tensor([a.shape[0] for a in afs], dtype=torch.int32, device=dev))
        d = _lib.IpcsDesc()
        d.B, d.NV, d.NT, d.NE, d.N2, d.NAF = B, NV, NT, NE, NV + NE, NAF
        d.mu = float(mu)
        for k, v in self.t.items():
            setattr(d, k, v.data_ptr())
        self.desc = d

    def probe_forces(self, u: torch.Tensor, p: torch.Tensor, stream=None):
        """u (B,F,N2,2), p (B,F,NV) -> drag, lift (B,F)."""
        F = u.shape[1]
        assert u.shape == (self.B, F, self.N2, 2) and p.shape == (self.B, F, self.cap["NV"]), (u.shape, p.shape)
        u, p = u.contiguous(), p.contiguous()
        drag = torch.empty((self.B, F), dtype=torch.float64, device=self.device)
        lift = torch.empty_like(drag)
        rc = self.lib.mdq_probe_forces(C.byref(self.desc), F, u.data_ptr(), p.data_ptr(), drag.data_ptr(),
                                       lift.data_ptr(), _lib.stream_ptr(stream))
        _lib.check(rc, "mdq_probe_forces")
        return drag, lift


class SnapshotInterpolator:
    """Source = the original (smoothed) mesh with S stored snapshots; targets = coarsened meshes."""

    def __init__(self, topo: MeshTopology, coords: np.ndarray, u_snap: torch.Tensor, p_snap: torch.Tensor,
                 device="cuda", grid=(224, 64)):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.topo = topo
        self.S = int(u_snap.shape[0])
        assert u_snap.shape == (self.S, topo.np2, 2) and p_snap.shape == (self.S, topo.nv)
        x = np.asarray(coords, np.float64)
        X = x[topo.cells]  # (nt,3,2)
        J00, J01 = X[:, 1, 0] - X[:, 0, 0], X[:, 2, 0] - X[:, 0, 0]
        J10, J11 = X[:, 1, 1] - X[:, 0, 1], X[:, 2, 1] - X[:, 0, 1]
        det = J00 * J11 - J01 * J10
        geom = np.stack([J11 / det, -J01 / det, -J10 / det, J00 / det, np.abs(det)])
        # uniform location grid: every bin lists (ascending) the cells whose bounding box touches it
        gnx, gny = grid
        lo, hi = x.min(axis=0), x.max(axis=0)
        hx, hy = (hi[0] - lo[0]) / gnx, (hi[1] - lo[1]) / gny
        bmin = np.floor((X.min(axis=1) - lo) / (hx, hy) - 1e-9).astype(int)
        bmax = np.floor((X.max(axis=1) - lo) / (hx, hy) + 1e-9).astype(int)
        bmin = np.clip(bmin, 0, (gnx - 1, gny - 1))
        bmax = np.clip(bmax, 0, (gnx - 1, gny - 1))
        lists = [[] for _ in range(gnx * gny)]
        for c in range(topo.nt):
            for gy in range(bmin[c, 1], bmax[c, 1] + 1):
                base = gy * gnx
                for gx in range(bmin[c, 0], bmax[c, 0] + 1):
                    lists[base + gx].append(c)
        cent = X.mean(axis=1)
        for bidx, l in enumerate(lists):
            if not l:  # bin inside the airfoil hole: nearest cell (extrapolation never really happens there)
                gy, gx = divmod(bidx, gnx)
                pc = lo + ((gx + 0.5) * hx, (gy + 0.5) * hy)
                l.append(int(np.argmin(((cent - pc) ** 2).sum(axis=1))))
        bin_ptr = np.zeros(gnx * gny + 1, np.int32)
        bin_ptr[1:] = np.cumsum([len(l) for l in lists])
        bin_cells = np.concatenate([np.asarray(l, np.int32) for l in lists])
        dev = self.device
        self.t = dict(src_coords=torch.from_numpy(x).to(dev),
                      src_cell_dofs=torch.from_numpy(np.ascontiguousarray(topo.cell_dofs.T, np.int32)).to(dev),
                      src_geom=torch.from_numpy(geom).to(dev), bin_ptr=torch.from_numpy(bin_ptr).to(dev),
                      bin_cells=torch.from_numpy(bin_cells).to(dev),
                      # one record per source cell for the point location: vertex 0 + Jinv (same doubles as src_coords / src_geom)
                      src_cellrec=torch.from_numpy(np.ascontiguousarray(np.stack(
                          [X[:, 0, 0], X[:, 0, 1], geom[0], geom[1], geom[2], geom[3]], axis=1))).to(dev),
                      src_u=u_snap.to(dev, torch.float64).contiguous(), src_p=p_snap.to(dev, torch.float64).contiguous())
        self.grid = (gnx, gny, float(lo[0]), float(lo[1]), 1.0 / hx, 1.0 / hy)

    def interpolate(self, topos: Sequence[MeshTopology], coords: Sequence[np.ndarray], NP=None, NP1=None, stream=None):
        """-> u (B,S,NP,2), p (B,S,NP1): snapshot values at every target's P2 dof points (vertices, then edge
        midpoints in the target's edge numbering) / P1 points (vertices)."""
        B = len(topos)
        NP = NP or max(t.np2 for t in topos)
        NP1 = NP1 or max(t.nv for t in topos)
        pts = np.zeros((B, NP, 2))
        for b, (t, x) in enumerate(zip(topos, coords)):
            pts[b, :t.np2] = t.dof_coords(x)
        dev = self.device
        t_pts = torch.from_numpy(pts).to(dev)
        npts = torch.tensor([t.np2 for t in topos], dtype=torch.int32, device=dev)
        np1 = torch.tensor([t.nv for t in topos], dtype=torch.int32, device=dev)
        out_u = torch.zeros((B, self.S, NP, 2), dtype=torch.float64, device=dev)
        out_p = torch.zeros((B, self.S, NP1), dtype=torch.float64, device=dev)
        d = _lib.InterpDesc()
        d.B, d.S, d.NP, d.NP1 = B, self.S, NP, NP1
        d.src_nv, d.src_nt, d.src_n2 = self.topo.nv, self.topo.nt, self.topo.np2
        d.gnx, d.gny, d.x0, d.y0, d.inv_hx, d.inv_hy = self.grid
        d.npts, d.np1, d.points = npts.data_ptr(), np1.data_ptr(), t_pts.data_ptr()
        for k, v in self.t.items():
            setattr(d, k, v.data_ptr())
        d.out_u, d.out_p, d.out_cell = out_u.data_ptr(), out_p.data_ptr(), None
        rc = self.lib.mdq_interpolate_snapshots(C.byref(d), _lib.stream_ptr(stream))
        _lib.check(rc, "mdq_interpolate_snapshots")
        return out_u, out_p


def remove_vertex_delaunay(coords: np.ndarray, boundary_vertices: np.ndarray, idx: int):
    """Env2DAirfoil.py:477-496: drop vertex `idx`, Delaunay-triangulate ALL remaining points (Qhull via
    scipy, exactly the reference's call), drop the simplices made of boundary vertices only.
    Returns (new_coords, new_cells); raises ValueError when Qhull cannot triangulate."""
    bv = np.array(boundary_vertices, copy=True)
    bv[bv > idx] -= 1
    keep = np.ones(coords.shape[0], dtype=bool)
    keep[idx] = False
    new_coords = coords[keep]
    tri = Delaunay(new_coords)
    cells = tri.simplices
    cells = cells[np.sum(np.isin(cells, bv), axis=1) != 3]
    return new_coords, cells


def polygon_distance(poly: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """`Polygon(poly).distance(Point(p))` for many points: 0 inside / on the ring, else the distance to the
    nearest ring segment (closing segment included)."""
    a = poly
    b = np.roll(poly, -1, axis=0)
    ab = b - a  # (m,2)
    ap = pts[:, None, :] - a[None, :, :]  # (n,m,2)
    t = np.clip((ap * ab[None]).sum(-1) / (ab * ab).sum(-1)[None], 0.0, 1.0)
    q = a[None] + t[..., None] * ab[None]
    dist = np.sqrt(((pts[:, None, :] - q) ** 2).sum(-1)).min(axis=1)
    x, y = pts[:, 0:1], pts[:, 1:2]
    x1, y1, x2, y2 = a[None, :, 0], a[None, :, 1], b[None, :, 0], b[None, :, 1]
    cond = (y1 > y) != (y2 > y)
    with np.errstate(divide="ignore", invalid="ignore"):
        xin = x1 + (y - y1) * (x2 - x1) / (y2 - y1)
    inside = (np.where(cond, x < xin, False).sum(axis=1) % 2) == 1
    dist[inside] = 0.0
    return dist


def red_refine(coords: np.ndarray, cells: np.ndarray):
    """Uniform (red) refinement: every triangle -> 4, new vertices at the edge midpoints (boundary
    midpoints stay on the straight boundary segments).  Used for the refined ~6k-triangle stress
    configuration of BASELINE.json."""
    topo = MeshTopology(coords, cells)
    nv = topo.nv
    mid = 0.5 * (topo.coords[topo.edges[:, 0]] + topo.coords[topo.edges[:, 1]])
    new_coords = np.concatenate([topo.coords, mid])
    c, ce = topo.cells, topo.cell_edges + nv  # local edge k is opposite local vertex k
    new_cells = np.concatenate([
        np.stack([c[:, 0], ce[:, 2], ce[:, 1]], axis=1),
        np.stack([c[:, 1], ce[:, 0], ce[:, 2]], axis=1),
        np.stack([c[:, 2], ce[:, 1], ce[:, 0]], axis=1),
        np.stack([ce[:, 0], ce[:, 1], ce[:, 2]], axis=1)])
    return new_coords, new_cells.astype(np.int32)


def remesh_batch(coords: np.ndarray, cells: np.ndarray, nv: np.ndarray, nt: np.ndarray, remove_idx: np.ndarray,
                 smooth_iters: int = 50, nthreads: int = 0) -> np.ndarray:
    """In-place batched vertex removal + Delaunay restoration + smoothing on the host engine
    (`mdq_remesh_host`).  coords (B,NV,2) f8, cells (B,NT,3) i4, nv/nt (B,) i4, remove_idx (B,) i4.
    Returns status (B,) (0 = ok)."""
    lib = _lib.load()
    B, NV = coords.shape[0], coords.shape[1]
    NT = cells.shape[1]
    assert coords.dtype == np.float64 and cells.dtype == np.int32 and coords.flags.c_contiguous and cells.flags.c_contiguous
    assert nv.dtype == np.int32 and nt.dtype == np.int32
    rem = np.ascontiguousarray(remove_idx, np.int32)
    status = np.zeros(B, np.int32)
    rc = lib.mdq_remesh_host(B, NV, NT, coords.ctypes.data, cells.ctypes.data, nv.ctypes.data, nt.ctypes.data,
                             rem.ctypes.data, int(smooth_iters), int(nthreads), status.ctypes.data)
    _lib.check(rc, "mdq_remesh_host")
    return status


class HostTopologyBatch:
    """Host arrays + descriptor of `mdq_env_topology_host` for B environments (capacities fixed at construction).

    `h` holds the outputs (ne, cell_dofs, points, naf, af_facets, nremovable, nsel, n_closest, coord_map, nedges,
    edge_src, edge_dst, edge_len); with `ipcs=True` also `hi`, the index data of the matrix-free IPCS path on every
    mesh (mdq_ipcs_topo_out: mf_scat, cell_outflow, bcu_flag, bcu_gx, bcp_flag, nbo, bo_*, g1_*, g2_*, sl1_*)."""

    def _zeros(self, name, shape, dtype):
        """Zeroed host array; page-locked (and registered under `name` for `upload`) when a GPU is present."""
        tdt = {np.float64: torch.float64, np.int32: torch.int32, np.int8: torch.int8, np.uint8: torch.uint8}[dtype]
        t = torch.zeros(shape, dtype=tdt, pin_memory=self._pin)
        self.pinned[name] = t
        return t.numpy()

    def upload(self, name, device):
        """Asynchronous H2D copy of a registered array on the current stream (the host engine does not touch the
        array again before the caller has synchronised on results of this step)."""
        return self.pinned[name].to(device, non_blocking=True)

    def __init__(self, B, NV, NT, NE, NAF, N, EMAX, polygon, ipcs=False, nbo_cap=64, nse1_cap=0):
        self.lib = _lib.load()
        NP = NV + NE
        self.B, self.NV, self.NT, self.NE, self.NP, self.NAF, self.N, self.EMAX = B, NV, NT, NE, NP, NAF, N, EMAX
        self.polygon = np.ascontiguousarray(polygon, dtype=np.float64)
        # page-locked host arrays (when a GPU is present): uploads run at full PCIe rate and asynchronously
        self._pin = torch.cuda.is_available()
        self.pinned = {}
        z = self._zeros
        self.coords = z("coords", (B, NV, 2), np.float64)
        self.cells = z("cells", (B, NT, 3), np.int32)
        self.nv = z("nv", (B,), np.int32)
        self.nt = z("nt", (B,), np.int32)
        self.offset = np.zeros(B, np.int32)
        self.h = dict(ne=z("ne", (B,), np.int32), cell_dofs=z("cell_dofs", (B, 6, NT), np.int32),
                      points=z("points", (B, NP, 2), np.float64), naf=z("naf", (B,), np.int32),
                      af_facets=z("af_facets", (B, NAF, 2), np.int32),
                      nremovable=np.zeros(B, np.int32), nsel=z("nsel", (B,), np.int32),
                      n_closest=z("n_closest", (B, N), np.int32),
                      coord_map=np.zeros((B, N), np.int32), nedges=np.zeros(B, np.int32),
                      edge_src=np.zeros((B, EMAX), np.int32), edge_dst=np.zeros((B, EMAX), np.int32),
                      edge_len=np.zeros((B, EMAX)))
        d = _lib.EnvTopoDesc()
        d.B, d.NV, d.NT, d.NP, d.NAF, d.N, d.EMAX, d.npoly = B, NV, NT, NP, NAF, N, EMAX, self.polygon.shape[0]
        d.coords, d.cells, d.nv, d.nt = (a.ctypes.data for a in (self.coords, self.cells, self.nv, self.nt))
        d.offset, d.polygon = self.offset.ctypes.data, self.polygon.ctypes.data
        for k, a in self.h.items():
            setattr(d, k, a.ctypes.data)
        self.hi = None
        if ipcs:
            NBO, NBE = int(nbo_cap), 6 * int(nbo_cap)
            NSE1 = int(nse1_cap) if nse1_cap else 64 * 16 * (NV // 64 + 1)
            self.NBO, self.NBE, self.NSE1 = NBO, NBE, NSE1
            self.hi = dict(mf_scat=z("mf_scat", (B, 6, NT), np.int32), cell_outflow=z("cell_outflow", (B, NT), np.int8),
                           bcu_flag=z("bcu_flag", (B, NP), np.uint8), bcu_gx=z("bcu_gx", (B, NP), np.float64),
                           bcp_flag=z("bcp_flag", (B, NV), np.uint8), nbo=z("nbo", (B,), np.int32),
                           bo_rows=z("bo_rows", (B, NBO), np.int32), bo_ptr=z("bo_ptr", (B, NBO + 1), np.int32),
                           bo_col=z("bo_col", (B, NBE), np.int32), bo_src=z("bo_src", (B, NBE), np.int32),
                           g1_ptr=z("g1_ptr", (B, NV + 1), np.int32), g1_src=z("g1_src", (B, 3 * NT), np.int32),
                           g2_ptr=z("g2_ptr", (B, NP + 1), np.int32), g2_src=z("g2_src", (B, 6 * NT), np.int32),
                           sl1_off=z("sl1_off", (B, NV // 64 + 2), np.int32), sl1_col=z("sl1_col", (B, NSE1), np.int32))
            o = _lib.IpcsTopoOut()
            o.NBO, o.NBE, o.NSE1 = NBO, NBE, NSE1
            for k, a in self.hi.items():
                setattr(o, k, a.ctypes.data)
            self._ipcs_out = o
            d.ipcs = C.cast(C.pointer(o), C.c_void_p)
        self.desc = d

    def run(self, nthreads=0):
        status = np.zeros(self.B, np.int32)
        _lib.check(self.lib.mdq_env_topology_host(C.byref(self.desc), int(nthreads), status.ctypes.data),
                   "mdq_env_topology_host")
        if (status != 0).any():
            raise _lib.MeshDQNHipError(f"topology engine failed: env {np.flatnonzero(status)} status {status[status != 0]}")
        return status


_SMOOTH_WS = {}
_WS = {}


def _workspace(kind: str, device, sp, nbytes: int, tag=()):
    """Caller-owned scratch of an entry point (`*_workspace_bytes`): one torch tensor per (kind, device, stream, sizes) -
    launches on one stream are ordered and may share it, launches on different streams get their own.  (0, 0) when the
    entry point needs none; raises when the sizes are beyond the kernels (< 0)."""
    if nbytes < 0:
        raise _lib.MeshDQNHipError(f"{kind}: mesh capacities beyond the kernels")
    if nbytes == 0:
        return None, 0
    key = (kind, torch.device(device), int(sp.value or 0)) + tuple(tag)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws.data_ptr(), ws.numel()


def smooth_batch_gpu(coords: torch.Tensor, cells: torch.Tensor, nv: torch.Tensor, nt: torch.Tensor,
                     iterations: torch.Tensor, stream=None, fast: bool = True) -> None:
    """In-place `mesh.smooth(n)` of B meshes on the GPU: coords (B,NV,2) f8, cells (B,NT,3) i4, nv / nt / iterations (B,) i4
    device tensors; iterations[b] = 0 leaves mesh b untouched.  `fast` (default): `mdq_smooth_fast` - three careful sweeps,
    then the full-step sweeps as blocked triangular solves, validated in parallel; `fast=False`: `mdq_smooth` (the
    per-vertex walk, every sweep), which the fast path is tested against."""
    lib = _lib.load()
    B, NV = coords.shape[0], coords.shape[1]
    NT = cells.shape[1]
    assert coords.dtype == torch.float64 and cells.dtype == torch.int32 and coords.is_contiguous() and cells.is_contiguous()
    assert nv.dtype == torch.int32 and nt.dtype == torch.int32 and iterations.dtype == torch.int32
    if not fast:
        sp = _lib.stream_ptr(stream)
        wp, wn = _workspace("smooth", coords.device, sp, int(lib.mdq_smooth_workspace_bytes(B, NV, NT)), (B, NV, NT))
        _lib.check(lib.mdq_smooth(B, NV, NT, coords.data_ptr(), cells.data_ptr(), nv.data_ptr(), nt.data_ptr(),
                                  iterations.data_ptr(), wp, wn, sp), "mdq_smooth")
        return
    # workspace (block inverses of every mesh): one per (device, stream, B, NV) - launches on one stream are ordered
    sp = _lib.stream_ptr(stream)
    key = (coords.device, int(sp.value or 0), B, NV)
    ws = _SMOOTH_WS.get(key)
    if ws is None:
        nbytes = int(lib.mdq_smooth_fast_workspace_bytes(B, NV))
        ws = _SMOOTH_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=coords.device)
    _lib.check(lib.mdq_smooth_fast(B, NV, NT, coords.data_ptr(), cells.data_ptr(), nv.data_ptr(), nt.data_ptr(),
                                   iterations.data_ptr(), ws.data_ptr(), ws.numel(), sp), "mdq_smooth_fast")


def smooth_env_gpu(coords: torch.Tensor, cells: torch.Tensor, nv: torch.Tensor, nt: torch.Tensor, rem: torch.Tensor,
                   rstat: torch.Tensor, iterations: int = 50, stream=None) -> None:
    """The smoothing of an env step (`mdq_smooth_fast_env`): `iterations` sweeps for the environments whose vertex removal
    succeeded (rem >= 0 and rstat == 0: the outputs of `mdq_env_act` / `mdq_remesh`), the others untouched."""
    lib = _lib.load()
    B, NV, NT = coords.shape[0], coords.shape[1], cells.shape[1]
    sp = _lib.stream_ptr(stream)
    key = (coords.device, int(sp.value or 0), B, NV)
    ws = _SMOOTH_WS.get(key)
    if ws is None:
        ws = _SMOOTH_WS[key] = torch.empty(int(lib.mdq_smooth_fast_workspace_bytes(B, NV)), dtype=torch.uint8, device=coords.device)
    _lib.check(lib.mdq_smooth_fast_env(B, NV, NT, coords.data_ptr(), cells.data_ptr(), nv.data_ptr(), nt.data_ptr(),
                                       rem.data_ptr(), rstat.data_ptr(), int(iterations), ws.data_ptr(), ws.numel(), sp),
               "mdq_smooth_fast_env")


def smooth_fast_stats(device, B: int, NV: int, stream=None) -> np.ndarray:
    """Diagnostics of the last `smooth_batch_gpu(fast=True)` launch with these sizes on this stream, (B, 4) int32 per
    environment: [sweeps handed back to the careful walk (0 = everything ran in the blocked solve), sweeps that needed repair
    rounds (vertices with limited steps), repair rounds in total, pipelined sweeps whose validation failed]."""
    ws = _SMOOTH_WS.get((torch.device(device) if not isinstance(device, torch.device) else device,
                         int(_lib.stream_ptr(stream).value or 0), B, NV))
    if ws is None and stream is None:       # (no stream given: the workspace of these sizes on whatever stream used it last)
        for key, val in _SMOOTH_WS.items():
            if key[2] == B and key[3] == NV:
                ws = val
    if ws is None:
        raise KeyError("no mdq_smooth_fast workspace for these sizes on this stream")
    # (the diagnostics are the tail of the workspace: block inverses + pair couplings, then [B] redo + [B][3] + 256 spare bytes)
    off = int(_lib.load().mdq_smooth_fast_workspace_bytes(B, NV)) - 16 * B - 256
    raw = ws[off:off + 16 * B].view(torch.int32).cpu().numpy()
    out = np.concatenate([raw[:B, None], raw[B:4 * B].reshape(B, 3)], axis=1)
    return out          # (column 3: pipelined sweeps redone checked | sweeps taken in plain index order << 16)


class DeviceTopologyBatch:
    """Device-resident counterpart of `HostTopologyBatch` (`mdq_env_topology`): inputs `coords` (B,NV,2), `cells`
    (B,NT,3), `nv`, `nt`, `offset` and all outputs are device tensors (`t` for the mesh / state outputs, `ti` for the
    matrix-free IPCS index data); nothing but what the caller asks for ever crosses PCIe."""

    def __init__(self, B, NV, NT, NE, NAF, N, EMAX, polygon, device, ipcs=False, nbo_cap=64, nse1_cap=0, flow_only=False):
        self.lib = _lib.load()
        dev = torch.device(device)
        NP = NV + NE
        self.B, self.NV, self.NT, self.NE, self.NP, self.NAF, self.N, self.EMAX = B, NV, NT, NE, NP, NAF, N, EMAX
        self.device = dev

        def z(shape, dt):
            return torch.zeros(shape, dtype=dt, device=dev)

        i32, f64 = torch.int32, torch.float64
        self.polygon = torch.as_tensor(np.ascontiguousarray(polygon, dtype=np.float64), device=dev)
        self.coords, self.cells = z((B, NV, 2), f64), z((B, NT, 3), i32)
        self.nv, self.nt, self.offset = z((B,), i32), z((B,), i32), z((B,), i32)
        self.status = z((B,), i32)
        self.t = dict(ne=z((B,), i32), cell_dofs=z((B, 6, NT), i32), points=z((B, NP, 2), f64), naf=z((B,), i32),
                      af_facets=z((B, NAF, 2), i32), nremovable=z((B,), i32), nsel=z((B,), i32),
                      n_closest=z((B, N), i32), coord_map=z((B, N), i32), nedges=z((B,), i32),
                      edge_src=z((B, EMAX), i32), edge_dst=z((B, EMAX), i32), edge_len=z((B, EMAX), f64))
        d = _lib.EnvTopoDesc()
        d.B, d.NV, d.NT, d.NP, d.NAF, d.N, d.EMAX, d.npoly = B, NV, NT, NP, NAF, N, EMAX, self.polygon.shape[0]
        d.coords, d.cells, d.nv, d.nt = (a.data_ptr() for a in (self.coords, self.cells, self.nv, self.nt))
        d.offset, d.polygon = self.offset.data_ptr(), self.polygon.data_ptr()
        for k, a in self.t.items():
            setattr(d, k, a.data_ptr())
        self.ti = None
        if ipcs:
            NBO, NBE = int(nbo_cap), 6 * int(nbo_cap)
            NSE1 = int(nse1_cap) if nse1_cap else 64 * 16 * (NV // 64 + 1)
            self.NBO, self.NBE, self.NSE1 = NBO, NBE, NSE1
            self.ti = dict(mf_scat=z((B, 6, NT), i32), cell_outflow=z((B, NT), torch.int8),
                           bcu_flag=z((B, NP), torch.uint8), bcu_gx=z((B, NP), f64), bcp_flag=z((B, NV), torch.uint8),
                           nbo=z((B,), i32), bo_rows=z((B, NBO), i32), bo_ptr=z((B, NBO + 1), i32),
                           bo_col=z((B, NBE), i32), bo_src=z((B, NBE), i32), g1_ptr=z((B, NV + 1), i32),
                           g1_src=z((B, 3 * NT), i32), g2_ptr=z((B, NP + 1), i32), g2_src=z((B, 6 * NT), i32),
                           sl1_off=z((B, NV // 64 + 2), i32), sl1_col=z((B, NSE1), i32))
            o = _lib.IpcsTopoOut()
            o.NBO, o.NBE, o.NSE1 = NBO, NBE, NSE1
            # an engine that only feeds the IPCS step (the flow stream's private engine): no selection, no state graph
            o.flow_only = 1 if flow_only else 0
            for k, a in self.ti.items():
                setattr(o, k, a.data_ptr())
            self._ipcs_out = o
            d.ipcs = C.cast(C.pointer(o), C.c_void_p)
        # the tables of the large-mesh kernel instance: this engine's own workspace (engines on different streams never share)
        nb = int(self.lib.mdq_env_topology_workspace_bytes(C.byref(d)))
        if nb < 0:
            raise _lib.MeshDQNHipError("mdq_env_topology: capacity above 16384 vertices / 32768 triangles / 65536 P2 dofs")
        self.workspace = torch.empty(nb, dtype=torch.uint8, device=dev) if nb else None
        d.workspace, d.workspace_bytes = (self.workspace.data_ptr() if nb else None), nb
        self.desc = d

    def take_edges_from(self, cell_dofs=None, ne=None):
        """A `flow_only` engine: take the edge numbering from the cell dofs (B,6,NT) / edge counts (B,) that another engine's
        run derived from the SAME meshes instead of finding it again through the hash table (None: number the edges here)."""
        if self.ti is None:
            raise ValueError("take_edges_from needs an engine with the IPCS index data")
        self._edges_in = (cell_dofs, ne)         # (kept alive)
        self._ipcs_out.cell_dofs_in = None if cell_dofs is None else cell_dofs.data_ptr()
        self._ipcs_out.ne_in = None if ne is None else ne.data_ptr()

    def set_handover(self, coords=None, cells=None, nv=None, nt=None, cell_dofs=None, ne=None):
        """Second set of outputs of the next `run()` calls (`mdq_topo_handover`): the kernel also writes the mesh and its
        edge numbering to these tensors (another engine's inputs); no arguments: off."""
        if coords is None:
            self._handover = None
            self.desc.handover = None
            return
        h = _lib.TopoHandover()
        for k, v in dict(coords=coords, cells=cells, nv=nv, nt=nt, cell_dofs=cell_dofs, ne=ne).items():
            want = getattr(self, k) if k in ("coords", "cells", "nv", "nt") else self.t[k]
            if v.shape != want.shape or v.dtype != want.dtype or not v.is_contiguous():
                raise ValueError(f"topology hand-over: `{k}` does not match the engine's array")
            setattr(h, k, v.data_ptr())
        self._handover = (h, coords, cells, nv, nt, cell_dofs, ne)     # (keeps the struct and the tensors alive)
        self.desc.handover = C.cast(C.pointer(h), C.c_void_p)

    def run(self, stream=None, check=True):
        _lib.check(self.lib.mdq_env_topology(C.byref(self.desc), _lib.stream_ptr(stream), self.status.data_ptr()),
                   "mdq_env_topology")
        if check:
            st = self.status.cpu().numpy()
            if (st != 0).any():
                raise _lib.MeshDQNHipError(f"topology kernel failed: env {np.flatnonzero(st)} status {st[st != 0]}")


def remesh_batch_gpu(coords: torch.Tensor, cells: torch.Tensor, nv: torch.Tensor, nt: torch.Tensor,
                     remove_idx: torch.Tensor, status: torch.Tensor, stream=None) -> None:
    """In-place batched vertex removal + Delaunay restoration on the GPU (`mdq_remesh`; no smoothing: follow with
    `smooth_batch_gpu`).  coords (B,NV,2) f8, cells (B,NT,3) i4, nv / nt / remove_idx / status (B,) i4 device tensors."""
    lib = _lib.load()
    B, NV, NT = coords.shape[0], coords.shape[1], cells.shape[1]
    assert coords.dtype == torch.float64 and cells.dtype == torch.int32 and coords.is_contiguous() and cells.is_contiguous()
    for a in (nv, nt, remove_idx, status):
        assert a.dtype == torch.int32 and a.is_cuda
    sp = _lib.stream_ptr(stream)
    wp, wn = remesh_workspace(coords.device, sp, B, NV, NT)
    _lib.check(lib.mdq_remesh(B, NV, NT, coords.data_ptr(), cells.data_ptr(), nv.data_ptr(), nt.data_ptr(),
                              remove_idx.data_ptr(), status.data_ptr(), wp, wn, sp), "mdq_remesh")


def remesh_workspace(device, sp, B: int, NV: int, NT: int):
    """(pointer, bytes) of the workspace of `mdq_remesh` / `mdq_remesh_act` for these capacities on this stream (None, 0 for
    meshes whose tables fit LDS)."""
    return _workspace("remesh", device, sp, int(_lib.load().mdq_remesh_workspace_bytes(B, NV, NT)), (B, NV, NT))
