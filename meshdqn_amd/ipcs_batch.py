"""Batched IPCS solver: B environments (meshes) advanced together on one MI355X.

Host side of `mdq_ipcs_assemble` / `mdq_ipcs_evolve` / `mdq_probe_forces`
(include/meshdqn_hip.h).  Builds the padded index arrays from `MeshTopology`,
owns the device tensors (torch = allocator + stream plumbing only) and hands
raw device pointers to the C ABI.  No CPU fallback.
"""
from __future__ import annotations

import os
import ctypes as C
from typing import Sequence

import numpy as np
import torch

from . import _lib
from .topology import conflict_free_cell_order, morton_cell_order, TAG_AIRFOIL, TAG_OUTFLOW, MeshTopology


def smooth_coords(topo: MeshTopology, iterations: int = 50, coords: np.ndarray | None = None) -> np.ndarray:
    """DOLFIN `Mesh.smooth(iterations)` (`flow_solver.py:65-67,236-237`) via the
    library's host routine; returns the new coordinates (topology unchanged)."""
    lib = _lib.load()
    x = np.array(topo.coords if coords is None else coords, dtype=np.float64, order="C", copy=True)
    nbr_ptr, nbr, vc_ptr, vc = topo.vertex_adjacency()
    cells = np.ascontiguousarray(topo.cells, dtype=np.int32)
    onb = np.ascontiguousarray(topo.on_boundary, dtype=np.uint8)
    nbr_ptr = np.ascontiguousarray(nbr_ptr, np.int64)
    nbr = np.ascontiguousarray(nbr, np.int64)
    vc_ptr = np.ascontiguousarray(vc_ptr, np.int64)
    vc = np.ascontiguousarray(vc, np.int64)
    rc = lib.mdq_smooth_host(x.ctypes.data, topo.nv, cells.ctypes.data, topo.nt, nbr_ptr.ctypes.data,
                             nbr.ctypes.data, vc_ptr.ctypes.data, vc.ctypes.data, onb.ctypes.data, int(iterations))
    _lib.check(rc, "mdq_smooth_host")
    return x


class IpcsBatch:
    """Device-resident batch of Taylor-Hood IPCS problems."""

    def __init__(self, topos: Sequence[MeshTopology], coords: Sequence[np.ndarray] | None = None,
                 mu: float = 1e-3, rho: float = 1.0, dt: float = 1e-3, rtol: float = 1e-10,
                 maxit=(200, 4000, 200), device: str | torch.device = "cuda", capacities: dict | None = None,
                 mode: int = -1, pressure_direct: bool = True, pressure_parts: int = 16,
                 cell_order: str = "auto", pcg_degree: int = 0):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MeshDQNHipError("IpcsBatch needs a GPU device (no CPU fallback)")
        self.topos = list(topos)
        B = len(self.topos)
        if B == 0:
            raise ValueError("empty batch")
        if cell_order == "auto":
            # the LDS-resident operator modes (every mesh of the batch within 3 584 velocity dofs): the conflict-free order of the
            # LDS-atomic mode 3; beyond them (element tiles, modes 5 / 7): a spatial order, so that a chunk of 1 024 triangles
            # shares its rows (2 200 touched rows per chunk on the red-refined ys930 instead of 5 100: the chunk's input rows fit
            # the kernels' LDS stage, and a row meets 1.06 chunks instead of 2.4)
            cell_order = "conflictfree" if max(t.np2 for t in self.topos) <= 3584 else "morton"
        if cell_order == "morton":
            cache = {}
            for t in self.topos:
                if id(t) not in cache:
                    cache[id(t)] = t.permuted(morton_cell_order(t.coords, t.cells))
            self.topos = [cache[id(t)] for t in self.topos]
        elif cell_order == "conflictfree":
            # internal cell order: the 64 cells a wave handles in one instruction share no dof, so the LDS atomics of
            # the matrix-free kernels never collide inside an instruction (dof numbering and all vectors unchanged)
            cache = {}
            for t in self.topos:
                if id(t) not in cache:
                    cache[id(t)] = t.permuted(conflict_free_cell_order(t.cells))
            self.topos = [cache[id(t)] for t in self.topos]
        elif cell_order != "mesh":
            raise ValueError("cell_order must be 'auto', 'conflictfree', 'morton' or 'mesh'")
        coords = [t.coords for t in self.topos] if coords is None else list(coords)
        self.B = B
        self.mu, self.rho, self.dt, self.rtol = float(mu), float(rho), float(dt), float(rtol)
        self.maxit = tuple(int(m) for m in maxit)

        per = [self._host_arrays(t, x) for t, x in zip(self.topos, coords)]
        cap = dict(NV=max(p["nv"] for p in per), NT=max(p["nt"] for p in per), NE=max(p["ne"] for p in per),
                   NNZ2=max(p["colidx2"].size for p in per), NNZ1=max(p["colidx1"].size for p in per),
                   NAF=max(max(p["af"].shape[0] for p in per), 1),
                   NSE2=max(p["sl2_col"].size for p in per), NSE1=max(p["sl1_col"].size for p in per),
                   NBO=max(max(p["bo_rows"].size for p in per), 1), NBE=max(max(p["bo_col"].size for p in per), 1))
        if capacities:
            for k, val in capacities.items():
                if val < cap[k]:
                    raise ValueError(f"capacity {k}={val} smaller than required {cap[k]}")
                cap[k] = int(val)
        self.cap = cap
        NV, NT, NE, NNZ2, NNZ1, NAF = (cap[k] for k in ("NV", "NT", "NE", "NNZ2", "NNZ1", "NAF"))
        NSE2, NSE1 = cap["NSE2"], cap["NSE1"]
        NBO, NBE = cap["NBO"], cap["NBE"]
        N2 = NV + NE
        self.N2 = N2

        def stack(key, shape, dtype, fill=0):
            out = np.full((B,) + shape, fill, dtype=dtype)
            for b, p in enumerate(per):
                a = p[key]
                out[(b,) + tuple(slice(0, s) for s in a.shape)] = a
            return out

        h = {}
        h["nv"] = np.array([p["nv"] for p in per], np.int32)
        h["nt"] = np.array([p["nt"] for p in per], np.int32)
        h["ne"] = np.array([p["ne"] for p in per], np.int32)
        h["naf"] = np.array([p["af"].shape[0] for p in per], np.int32)
        h["coords"] = stack("coords", (NV, 2), np.float64)
        h["cell_dofs"] = stack("cell_dofs_soa", (6, NT), np.int32)
        h["cell_outflow"] = stack("cell_outflow", (NT,), np.int8, fill=-1)
        h["rowptr2"] = stack("rowptr2", (N2 + 1,), np.int32)
        h["colidx2"] = stack("colidx2", (NNZ2,), np.int32)
        h["asm2_ptr"] = stack("asm2_ptr", (NNZ2 + 1,), np.int32)
        h["asm2_src"] = stack("asm2_src", (36 * NT,), np.int32)
        h["rowptr1"] = stack("rowptr1", (NV + 1,), np.int32)
        h["colidx1"] = stack("colidx1", (NNZ1,), np.int32)
        h["asm1_ptr"] = stack("asm1_ptr", (NNZ1 + 1,), np.int32)
        h["asm1_src"] = stack("asm1_src", (9 * NT,), np.int32)
        h["sl2_off"] = stack("sl2_off", (N2 // 64 + 2,), np.int32)
        h["sl2_col"] = stack("sl2_col", (NSE2,), np.int32)
        h["sl1_off"] = stack("sl1_off", (NV // 64 + 2,), np.int32)
        h["sl1_col"] = stack("sl1_col", (NSE1,), np.int32)
        NCH = (NT + 1023) // 1024
        if N2 > 4096:
            # capacity beyond the packed words (dof ids of 12 bits): the kernels read plain tile positions for EVERY mesh of
            # the batch (mode 5: element tiles with global vectors), also for the ones that would fit the packed form
            for t_, p_ in zip(self.topos, per):
                if t_.np2 <= 4096:
                    p_["mf_scat"], p_["mf_tptr"] = t_.matfree_maps(1024)
            # per chunk the rows its triangles touch (mode 5's row phase: ~2 500 of the refined mesh's 12 924 rows per chunk)
            for p_ in per:
                tp = p_["mf_tptr"]
                cnt = np.diff(tp, axis=1)
                p_["mf_rcnt"] = (cnt > 0).sum(axis=1).astype(np.int32)
            NRL = int(max(p_["mf_rcnt"].max() for p_ in per))
            # bit 31 / 30 of the row word: this chunk is the FIRST / LAST one that touches the row - the row phase starts the
            # row's sum at 0 instead of reading a zero-filled accumulation vector (the last-touch bit is used by an experiment
            # build only, see tile_apply_global); valid when every row of every mesh is touched by some chunk (every dof
            # belongs to a cell)
            flags_ok = os.environ.get("MDQ_NO_RL_FLAGS", "") != "1"      # (A / B switch: plain row words, rl_flags = 0)
            for t_, p_ in zip(self.topos, per):
                flags_ok = flags_ok and bool((np.diff(p_["mf_tptr"], axis=1) > 0)[:, :t_.np2].any(axis=0).all())
            for p_ in per:
                tp = p_["mf_tptr"]
                cnt = np.diff(tp, axis=1)
                touched = cnt > 0
                first = touched & (np.cumsum(touched, axis=0) == 1)
                last = touched & (np.cumsum(touched[::-1], axis=0)[::-1] == 1)
                rl = np.zeros((tp.shape[0], NRL, 2), np.int32)
                for c in range(tp.shape[0]):
                    rows = np.flatnonzero(touched[c])
                    word = rows.astype(np.uint32)
                    if flags_ok:
                        word = word | (first[c, rows].astype(np.uint32) << 31) | (last[c, rows].astype(np.uint32) << 30)
                    rl[c, :rows.size, 0] = word.view(np.int32)
                    rl[c, :rows.size, 1] = tp[c, rows] | (cnt[c, rows] << 16)
                p_["mf_rlist"] = rl
                # packed local maps (round 5): position of every dof of a triangle in its chunk's row list | tile position << 16 -
                # the kernel stages the chunk's input rows in LDS through the row list and gathers from there
                # (MDQ_NO_LPOS=1: the per-triangle gathers from global memory of round 4, A / B switch)
                if NRL <= 0xFFFF and os.environ.get("MDQ_NO_LPOS", "") != "1":
                    cdofs = p_["cell_dofs_soa"]                       # (6, nt)
                    scat = p_["mf_scat"]
                    lpos = np.zeros_like(scat)
                    for c in range(tp.shape[0]):
                        lo_, hi_ = c * 1024, min((c + 1) * 1024, cdofs.shape[1])
                        rows = np.flatnonzero(touched[c])
                        loc = np.searchsorted(rows, cdofs[:, lo_:hi_])
                        assert (rows[loc] == cdofs[:, lo_:hi_]).all() and scat[:, lo_:hi_].max(initial=0) < 0x8000
                        lpos[:, lo_:hi_] = loc | (scat[:, lo_:hi_] << 16)
                    p_["mf_lpos"] = lpos.astype(np.int32)
            self._rl_flags = 1 if flags_ok else 0
            h["mf_rlist"] = stack("mf_rlist", (NCH, NRL, 2), np.int32)
            h["mf_rcnt"] = stack("mf_rcnt", (NCH,), np.int32)
            if all("mf_lpos" in p_ for p_ in per):
                h["mf_lpos"] = stack("mf_lpos", (6, NT), np.int32)
            self._NRL = NRL
        h["mf_scat"] = stack("mf_scat", (6, NT), np.int32)
        h["mf_tptr"] = stack("mf_tptr", (NCH, N2 + 1), np.int32)
        h["nbo"] = np.array([p["bo_rows"].size for p in per], np.int32)
        h["bo_rows"] = stack("bo_rows", (NBO,), np.int32)
        h["bo_ptr"] = stack("bo_ptr", (NBO + 1,), np.int32)
        h["bo_col"] = stack("bo_col", (NBE,), np.int32)
        h["bo_src"] = stack("bo_src", (NBE,), np.int32)
        h["g2_ptr"] = stack("g2_ptr", (N2 + 1,), np.int32)
        h["g2_src"] = stack("g2_src", (6 * NT,), np.int32)
        h["g1_ptr"] = stack("g1_ptr", (NV + 1,), np.int32)
        h["g1_src"] = stack("g1_src", (3 * NT,), np.int32)
        h["bcu_flag"] = stack("bcu_flag", (N2,), np.uint8)
        h["bcu_gx"] = stack("bcu_gx", (N2,), np.float64)
        h["bcp_flag"] = stack("bcp_flag", (NV,), np.uint8)
        h["af_facets"] = stack("af", (NAF, 2), np.int32)
        self.host = h
        self.per = per
        dev = self.device
        self.t = {k: torch.from_numpy(v).to(dev) for k, v in h.items()}

        def z(*shape):
            return torch.zeros(shape, dtype=torch.float64, device=dev)

        t = self.t
        t["geom"] = z(B, 5, NT)
        t["A1"] = z(B, NSE2, 4)
        t["Ms"] = z(B, NSE2)
        t["K1s"] = z(B, NSE1)
        t["lift1"] = z(B, N2, 2)
        t["lift3"] = z(B, N2, 2)
        t["idiag1"] = z(B, N2, 2)
        t["sdiagM"] = z(B, N2)
        t["sdiagK"] = z(B, NV)
        t["bo_val"] = z(B, NBE, 4)
        t["u_n"] = z(B, N2, 2)
        t["p_n"] = z(B, NV)
        nwork = int(self.lib.mdq_ipcs_workspace_doubles(B, NV, NT, NE))
        t["work"] = z(nwork)
        self.iters = torch.zeros((B, 3), dtype=torch.int32, device=dev)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)       # sticky (mdq_ipcs_desc.status, ABI 7): see `check`
        self.steps_done = 0

        d = _lib.IpcsDesc()
        d.B, d.NV, d.NT, d.NE, d.N2, d.NNZ2, d.NNZ1, d.NAF = B, NV, NT, NE, N2, NNZ2, NNZ1, NAF
        d.NSE2, d.NSE1 = NSE2, NSE1
        d.NBO, d.NBE = NBO, NBE
        d.mu, d.rho, d.dt, d.rtol = self.mu, self.rho, self.dt, self.rtol
        d.maxit_u, d.maxit_p, d.maxit_m = self.maxit
        if int(mode) in (-1, 3) and max(p["bo_max_per_thread"] for p in per) > 2:
            mode = -2 if int(mode) == -1 else mode      # (auto without the LDS-atomic mode 3)
            if int(mode) == 3:
                raise ValueError("mode 3 supports at most 2 outflow rows per row-owner thread")
        d.mode = int(mode)
        for name, _typ in _lib.IpcsDesc._fields_:
            if name in t:
                setattr(d, name, t[name].data_ptr())
        if os.environ.get("MDQ_NO_TILE_MAPS", "") == "1":     # (A / B: modes 5 / 7 through the dof <- slot lists, as on device-built index data)
            d.mf_tptr = d.mf_scat = d.mf_rlist = d.mf_rcnt = d.mf_lpos = None
        d.work_doubles = nwork
        d.status = self.status.data_ptr()
        d.NRL = getattr(self, "_NRL", 0)
        d.rl_flags = getattr(self, "_rl_flags", 0)
        d.pd_enabled = 0
        # Krylov pressure solve of mode 3: degree of the Chebyshev polynomial preconditioner (0, default: the plain Jacobi-CG
        # kernel).  Measured on ys930 (tools/time_pcg.py): iterations 154 -> 95 / 67 / 52 / 37 / 30 for degree 2 / 3 / 4 / 6 / 8,
        # kernel time unchanged (251-283 us against 253 us): the solve is bound by the LDS gathers of the operator
        # application (~1.6 us each, bank conflicts of the random vertex gather), not by its reductions, and the polynomial
        # trades the one for the other.  pcg_degree < 0: two-level additive preconditioner (8 x 7 geometric aggregates, coarse
        # matrix inverted in LDS): 154 -> 86 iterations, but 465 us per solve in its first version (aggregation + inversion
        # ~0.1 ms, two more barriers and a coarse product per iteration).  Both kept as options (the reference's Krylov option
        # is CG + AMG, flow_solver.py:152-155); the default is the Jacobi-CG.  Meshes beyond mode 3 (pressure vectors in LDS,
        # modes 0 / 4 / 5 / 7): 0 = auto, the two-level preconditioner with an O(n) aggregation from 2048 vertices on
        # (cg_pressure_2l_lds: 325 -> 158 iterations on the refined ys930), > 0 = Jacobi-CG, < 0 = two-level wherever it fits.
        d.pcg_degree = int(pcg_degree)
        self.desc = d
        self.assembled = False
        # True: substructuring factors built on the host (numpy, pressure_direct.py); "device": built by
        # mdq_ipcs_factorize_pressure (one workgroup per environment, no host work); False: Jacobi-CG
        self.pressure_direct = "device" if pressure_direct == "device" else bool(pressure_direct)
        self.pressure_parts = int(pressure_parts)
        self._pd_cache = {}

    # ------------------------------------------------------------------
    @staticmethod
    def _host_arrays(topo: MeshTopology, coords: np.ndarray) -> dict:
        bc = topo.boundary_conditions(coords)
        pat = topo.patterns()
        gat = topo.dof_gathers()
        rowptr2, colidx2, asm2_ptr, asm2_src = pat["p2"]
        rowptr1, colidx1, asm1_ptr, asm1_src = pat["p1"]
        out_f, _ = topo.facets(bc["tags"], TAG_OUTFLOW)
        af, af_edges = topo.facets(bc["tags"], TAG_AIRFOIL)
        sl2_off, sl2_col, pos2 = topo.sell_layout(rowptr2, colidx2)
        sl1_off, sl1_col, pos1 = topo.sell_layout(rowptr1, colidx1)
        cell_outflow = np.full(topo.nt, -1, dtype=np.int8)
        cell_outflow[out_f[:, 0]] = out_f[:, 1]
        # outflow-facet term as a row list: rows = the 3 P2 dofs of every outflow facet, columns = the 6
        # dofs of the adjacent cell (duplicates across neighbouring facets are simply summed by the owner)
        a_l = np.array([1, 0, 0])
        b_l = np.array([2, 2, 1])
        ent = []
        for c, k in out_f:
            for i in (a_l[k], b_l[k], 3 + k):
                for j in range(6):
                    ent.append((int(topo.cell_dofs[c, i]), int(topo.cell_dofs[c, j]), int(c) * 36 + int(i) * 6 + j))
        ent.sort()
        bo_rows = sorted({e[0] for e in ent})
        bo_ptr = np.zeros(len(bo_rows) + 1, np.int32)
        for e in ent:
            bo_ptr[bo_rows.index(e[0]) + 1] += 1
        bo_ptr = np.cumsum(bo_ptr).astype(np.int32)
        bo_col = np.array([e[1] for e in ent], np.int32)
        bo_src = np.array([e[2] for e in ent], np.int32)
        bo_rows = np.array(bo_rows, np.int32)
        bo_max_per_thread = int(np.bincount(bo_rows % 512, minlength=512).max()) if bo_rows.size else 0
        if topo.np2 <= 4096:
            mf_scat, mf_tptr = topo.matfree_packed(cell_outflow, 1024)
        else:  # matrix-free mode unavailable for this size (kernel falls back to the SELL operators)
            mf_scat, mf_tptr = topo.matfree_maps(1024)
        return dict(nv=topo.nv, nt=topo.nt, ne=topo.ne, coords=np.asarray(coords, np.float64),
                    cell_dofs_soa=np.ascontiguousarray(topo.cell_dofs.T, dtype=np.int32),
                    cell_outflow=cell_outflow,
                    rowptr2=rowptr2, colidx2=colidx2, asm2_ptr=asm2_ptr, asm2_src=asm2_src,
                    rowptr1=rowptr1, colidx1=colidx1, asm1_ptr=asm1_ptr, asm1_src=asm1_src,
                    sl2_off=sl2_off, sl2_col=sl2_col, sl1_off=sl1_off, sl1_col=sl1_col, pos2=pos2, pos1=pos1,
                    bo_max_per_thread=bo_max_per_thread, mf_scat=mf_scat, mf_tptr=mf_tptr, bo_rows=bo_rows, bo_ptr=bo_ptr, bo_col=bo_col, bo_src=bo_src,
                    g2_ptr=gat["p2"][0], g2_src=gat["p2"][1], g1_ptr=gat["p1"][0], g1_src=gat["p1"][1],
                    bcu_flag=bc["bcu_flag"], bcu_gx=bc["bcu_gx"], bcp_flag=bc["bcp_flag"],
                    af=af, af_edges=af_edges, tags=bc["tags"])

    # ------------------------------------------------------------------
    @property
    def u_n(self) -> torch.Tensor:
        """(B, N2, 2) velocity dofs [dof][component]."""
        return self.t["u_n"]

    @property
    def p_n(self) -> torch.Tensor:
        return self.t["p_n"]

    def reset_state(self):
        self.t["u_n"].zero_()
        self.t["p_n"].zero_()
        self.t["work"].zero_()  # (also drops the initial-guess history of the velocity solve)
        self.iters.zero_()
        self.steps_done = 0

    def assemble(self, stream=None):
        rc = self.lib.mdq_ipcs_assemble(C.byref(self.desc), _lib.stream_ptr(stream))
        _lib.check(rc, "mdq_ipcs_assemble")
        self.assembled = True
        if self.pressure_direct == "device":
            self.factorize_pressure_device(stream)
        elif self.pressure_direct:
            self._factorize_pressure()

    # capacities of the device-built factors (limits of mdq_ipcs_factorize_pressure: 8 parts, 112 interior nodes per
    # part, 48 separator nodes per part, 112 separator nodes)
    PD_DEVICE_CAP = dict(NPART=8, NPW=8 * 112 * 112, NPF=8 * 112 * 48, NPGI=8 * 48, NPS=112 * 112 + 8 * 48 * 48, NPGK=4096)

    def factorize_pressure_device(self, stream=None):
        """The factorisation of every environment's pressure matrix ON THE DEVICE (`mdq_ipcs_factorize_pressure`: the
        reference re-factorises with MUMPS after every remesh, flow_solver.py:318-328): fills the pd_* arrays the
        solve phase reads and switches the descriptor to the direct solve.  Needs the operators' set-up data (after
        `assemble()` or `mdq_ipcs_setup_matfree`).  `self.pd_status` (device int32 [B]): 0, or < 0 where a mesh exceeds
        the kernel's limits (that environment keeps the Krylov solve)."""
        d, B, NV, dev = self.desc, self.B, self.cap["NV"], self.device
        if getattr(self, "_pd_dev", None) is None:
            cap = self.PD_DEVICE_CAP
            i32, f64 = torch.int32, torch.float64
            z = lambda n, dt: torch.zeros((B, n), dtype=dt, device=dev)   # noqa: E731
            self._pd_dev = dict(pd_hdr=z(4, i32), pd_node=z(NV, i32), pd_meta=z(cap["NPART"] * 6, i32), pd_rowblk=z(NV, i32),
                                pd_W=z(cap["NPW"], f64), pd_F=z(cap["NPF"], f64), pd_gidx=z(cap["NPGI"], i32),
                                pd_Sinv=z(cap["NPS"], f64), pd_gk_ptr=z(NV + 1, i32), pd_gk_col=z(cap["NPGK"], i32),
                                pd_gk_val=z(cap["NPGK"], f64))
            self.pd_status = torch.zeros(B, dtype=i32, device=dev)
        for k, a in self._pd_dev.items():
            self.t[k] = a
            setattr(d, k, a.data_ptr())
        cap = self.PD_DEVICE_CAP
        d.NPART, d.NPW, d.NPF, d.NPGI, d.NPS, d.NPGK = (cap[k] for k in ("NPART", "NPW", "NPF", "NPGI", "NPS", "NPGK"))
        _lib.check(self.lib.mdq_ipcs_factorize_pressure(C.byref(d), self.pd_status.data_ptr(), _lib.stream_ptr(stream)),
                   "mdq_ipcs_factorize_pressure")
        d.pd_enabled = 1
        self.pds = None

    def update_inflow(self, profile, time: float, stream=None):
        """Time dependent inflow (flow_solver.py:70-73,369-371: `inflow.time = gtime` before the three assemblies apply
        the boundary conditions): new Dirichlet values profile(x, y, t) at the inlet dofs of every environment, then the
        vectors that depend on them (the lifting vectors A1[:, bc] g and M[:, bc] g) rebuilt on the device by
        `mdq_ipcs_setup_matfree` (which leaves the operators' other data as they are: same mesh)."""
        if not hasattr(self, "_inlet"):
            self._inlet = []
            for t_, p_ in zip(self.topos, self.per):
                d = t_.boundary_conditions(p_["coords"])["inlet_dofs"]
                self._inlet.append((d, t_.dof_coords(p_["coords"])[d]))
            self._gx_host = self.t["bcu_gx"].cpu().numpy().copy()
        for b, (d, xy) in enumerate(self._inlet):
            self._gx_host[b, d] = np.asarray(profile(xy[:, 0], xy[:, 1], float(time)), dtype=np.float64)
        self.t["bcu_gx"].copy_(torch.from_numpy(self._gx_host), non_blocking=True)
        rc = self.lib.mdq_ipcs_setup_matfree(C.byref(self.desc), _lib.stream_ptr(stream))
        _lib.check(rc, "mdq_ipcs_setup_matfree")

    def _factorize_pressure(self):
        """Host factorisation of every environment's (scaled, BC-eliminated) pressure matrix
        (`LUSolver('mumps')` of A2, flow_solver.py:150-159) -> substructuring factors on the device."""
        from .pressure_direct import build_pressure_direct
        torch.cuda.synchronize(self.device)
        K1s = self.t["K1s"].cpu().numpy()
        pds = []
        for b, p in enumerate(self.per):
            nv = p["nv"]
            key = (id(self.topos[b]), p["coords"].tobytes())
            if key not in self._pd_cache:
                rp, ci, pos = p["rowptr1"], p["colidx1"], p["pos1"]
                rows = np.repeat(np.arange(nv), np.diff(rp))
                K = np.zeros((nv, nv))
                K[rows, ci[:rp[-1]]] = K1s[b][pos]
                self._pd_cache[key] = build_pressure_direct(p["coords"], K, self.pressure_parts)
            pds.append(self._pd_cache[key])
        B, NV = self.B, self.cap["NV"]
        cap = dict(NPART=max(q["nparts"] for q in pds), NPW=max(max(q["W"].size for q in pds), 1),
                   NPF=max(max(q["F"].size for q in pds), 1), NPGI=max(max(q["gidx"].size for q in pds), 1),
                   NPS=max(max(q["Sinv"].size for q in pds), 1), NPGK=max(max(q["gk_col"].size for q in pds), 1))
        self.pd_cap = cap

        def stack(key, n, dtype):
            out = np.zeros((B, n), dtype=dtype)
            for b, q in enumerate(pds):
                a = q[key]
                out[b, :a.size] = a
            return out

        h = dict(pd_hdr=np.array([[q["nI"], q["nG"], q["nparts"], 0] for q in pds], np.int32),
                 pd_node=stack("node", NV, np.int32), pd_meta=stack("meta", cap["NPART"] * 6, np.int32),
                 pd_rowblk=stack("rowblk", NV, np.int32), pd_W=stack("W", cap["NPW"], np.float64),
                 pd_F=stack("F", cap["NPF"], np.float64), pd_gidx=stack("gidx", cap["NPGI"], np.int32),
                 pd_Sinv=stack("Sinv", cap["NPS"], np.float64), pd_gk_ptr=stack("gk_ptr", NV + 1, np.int32),
                 pd_gk_col=stack("gk_col", cap["NPGK"], np.int32), pd_gk_val=stack("gk_val", cap["NPGK"], np.float64))
        d = self.desc
        for k, a in h.items():
            self.t[k] = torch.from_numpy(a).to(self.device)
            setattr(d, k, self.t[k].data_ptr())
        d.NPART, d.NPW, d.NPF, d.NPGI, d.NPS, d.NPGK = (cap[k] for k in ("NPART", "NPW", "NPF", "NPGI", "NPS", "NPGK"))
        d.pd_enabled = 1
        self.pds = pds

    def evolve(self, nsteps: int = 1, stream=None, out=None):
        """Advance all environments `nsteps` IPCS steps; returns (drag, lift) (B,nsteps) device tensors."""
        if not self.assembled:
            self.assemble(stream)
        if out is None:
            drag = torch.empty((self.B, nsteps), dtype=torch.float64, device=self.device)
            lift = torch.empty_like(drag)
        else:
            drag, lift = out
        rc = self.lib.mdq_ipcs_evolve(C.byref(self.desc), int(nsteps), drag.data_ptr(), lift.data_ptr(),
                                      self.iters.data_ptr(), _lib.stream_ptr(stream))
        _lib.check(rc, "mdq_ipcs_evolve")
        self.steps_done += nsteps
        return drag, lift

    def check(self):
        """Synchronises and raises if any `evolve` since the last call abandoned a step (status words of the descriptor: the
        two-workgroup operator modes 4 / 7 give a step up when a team barrier times out - csrc/mdq_ipcs.hip `team_failed`; the
        forces of that step are NaN and u_n / p_n keep the last completed step).  `evolve` itself never synchronises."""
        st = self.status.cpu().numpy()
        if (st != 0).any():
            self.status.zero_()
            raise _lib.MeshDQNHipError(f"IPCS step abandoned in environment(s) {np.flatnonzero(st)[:8].tolist()}: team barrier "
                                       "time-out (another process or stream held CUs of a two-workgroup operator mode)")

    def evolve_timed(self, nsteps: int = 1, stream=None, out=None):
        """`evolve` with HIP events around every kernel (mode 3): returns (drag, lift, ms) where ms[3] are the
        accumulated durations of the velocity / pressure / correction kernels over the nsteps."""
        if not self.assembled:
            self.assemble(stream)
        if out is None:
            drag = torch.empty((self.B, nsteps), dtype=torch.float64, device=self.device)
            lift = torch.empty_like(drag)
        else:
            drag, lift = out
        ms = (C.c_double * 3)()
        rc = self.lib.mdq_ipcs_evolve_timed(C.byref(self.desc), int(nsteps), drag.data_ptr(), lift.data_ptr(),
                                            self.iters.data_ptr(), _lib.stream_ptr(stream), ms)
        _lib.check(rc, "mdq_ipcs_evolve_timed")
        self.steps_done += nsteps
        return drag, lift, [ms[0], ms[1], ms[2]]

    def probe_forces(self, u: torch.Tensor, p: torch.Tensor, stream=None):
        """u (B,F,N2,2), p (B,F,NV) -> drag, lift (B,F)."""
        F = u.shape[1]
        assert u.shape == (self.B, F, self.N2, 2) and p.shape == (self.B, F, self.cap["NV"])
        u = u.contiguous()
        p = p.contiguous()
        drag = torch.empty((self.B, F), dtype=torch.float64, device=self.device)
        lift = torch.empty_like(drag)
        rc = self.lib.mdq_probe_forces(C.byref(self.desc), F, u.data_ptr(), p.data_ptr(), drag.data_ptr(),
                                       lift.data_ptr(), _lib.stream_ptr(stream))
        _lib.check(rc, "mdq_probe_forces")
        return drag, lift

    # ------------------------------------------------------------------
    def algorithmic_bytes_per_step(self, iters_per_step=None) -> float:
        """SURVEY.md 8(d) accounting (assembled-CSR convention): every SpMV streams its CSR matrix
        once (fp64 values, int32 indices), vector passes 8 B per entry per vector.  This is the
        figure of merit the survey defines; the matrix-free modes move far fewer bytes
        (see `implemented_bytes_per_step`)."""
        tot = 0.0
        it = iters_per_step
        for b, p in enumerate(self.per):
            nv, nt, ne = p["nv"], p["nt"], p["ne"]
            n2 = nv + ne
            nnz2, nnz1 = p["colidx2"].size, p["colidx1"].size

            def spmv(nnz, n):
                return 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n
            a1 = spmv(4 * nnz2, 2 * n2)
            mm = spmv(nnz2, n2) * 2
            k1 = spmv(nnz1, nv)
            elem = 268.0 * nt
            iu, ip, im = (it[b] if it is not None else (6.0, 134.0, 3.0))
            tot += (3 * elem + a1 + iu * (2 * a1 + 10 * 8 * 2 * n2) + k1 + ip * (k1 + 6 * 8 * nv)
                    + mm + im * (mm + 6 * 8 * 2 * n2) + 200.0 * p["af"].shape[0])
        return tot

    def implemented_bytes_per_step(self, iters_per_step, mode: int = 3) -> float:
        """Global-memory bytes the IMPLEMENTED algorithm has to move per batch step (no credit for
        L2 hits; LDS / register resident data not counted), modes 2/3 (matrix-free) with the direct
        pressure solver:
          per operator application: element metadata 6 packed words + 5 geometry doubles per triangle
          per BiCGStab iteration (mode 3): x read + write (16 B per dof)
          per step: u_n, p_n read by the element loops and rewritten, u*, right-hand-side scratch of
          step 2, boundary data, direct-solver factors W, F, Sinv, K[G,I]."""
        tot = 0.0
        for b, p in enumerate(self.per):
            nv, nt, ne = p["nv"], p["nt"], p["ne"]
            n2 = nv + ne
            iu, ip, im = iters_per_step[b]
            apply_b = (24.0 + 40.0) * nt
            napply = 2 + 2 * iu + 2 + im  # rhs1, A x0, 2 per BiCGStab it, rhs3, M x0, 1 per CG it
            per_it_vec = 32.0 * n2 * iu
            state = (16.0 * n2 * 6 + 8.0 * nv * 6) + 12.0 * nt * 3 * 2 + 25.0 * n2
            if getattr(self, "pds", None) is not None and self.desc.pd_enabled:
                q = self.pds[b]
                prs = 8.0 * (q["W"].size + q["F"].size + q["Sinv"].size + q["gk_val"].size) + 4.0 * (
                    q["gk_col"].size + 2 * nv + q["gidx"].size)
            else:
                prs = ip * (12.0 * p["colidx1"].size)
            tot += napply * apply_b + per_it_vec + state + prs
        return tot

    def tile_mode_bytes_per_step(self, iters_per_step) -> float:
        """Global-memory bytes the element-tile modes (5 / 7) have to move per batch step, no cache credit - the ALGORITHMIC
        bytes of `evolve_team_tiles_kernel` / `evolve_kernel<5>` for the roofline of BASELINE configs[4]:
          per operator application and triangle: 6 packed local-map words + 5 geometry doubles + the outflow byte (65 B);
          per application and touched row of a chunk (T = sum of the chunks' row lists): the list entry (8 B) + the staged input row (16 B);
          per application and row: the accumulated result written and read once (32 B);
          per BiCGStab iteration (two applications): 15 vector streams of 16 B per row (p, s, x, r updates and the two
          epilogues' operands); per mass-CG iteration (one application): 8 streams;
          per step: right-hand-side element loops (3 x 65 B per triangle + u_n / p_n gathers 112 B per triangle), state vectors
          (u_n, p_n, u*, the history slots: 10 x 16 B per row), the pressure solve's factors (direct) or its matrix once (Krylov
          with the matrix on the chip)."""
        tot = 0.0
        for b, p in enumerate(self.per):
            nv, nt, ne = p["nv"], p["nt"], p["ne"]
            n2 = nv + ne
            iu, ip, im = iters_per_step[b]
            touched = float(np.asarray(p["mf_rcnt"]).sum()) if "mf_rcnt" in p else float(6 * nt)
            napply = 1 + 2 * iu + 1 + im                     # A x0, 2 per BiCGStab iteration, M x0, 1 per CG iteration
            apply_b = 65.0 * nt + 24.0 * touched + 32.0 * n2
            vec = (15.0 * iu + 8.0 * im) * 16.0 * n2
            rhs = 3 * (65.0 + 112.0) * nt
            state = 10 * 16.0 * n2 + 4 * 8.0 * nv
            if getattr(self, "pds", None) is not None and self.desc.pd_enabled:
                q = self.pds[b]
                prs = 8.0 * (q["W"].size + q["F"].size + q["Sinv"].size + q["gk_val"].size) + 4.0 * (
                    q["gk_col"].size + 2 * nv + q["gidx"].size)
            else:
                prs = 12.0 * p["colidx1"].size
            tot += napply * apply_b + vec + rhs + state + prs
        return tot

    def velocity_kernel_bytes(self, iters_per_step) -> float:
        """Global-memory bytes of ONE launch of the dominant kernel (at_velocity_kernel, mode 3) for the whole
        batch, no cache credit: per operator application 64 B of triangle metadata (6 packed words + 5 geometry
        doubles), 2 + 2*it applications (rhs1, A x0, two per BiCGStab iteration); x read + written per iteration
        (32 B per dof); per launch and dof: u_n (staged in LDS once), u* and its four predecessors read, five history
        slots written (the shift + the initial guess), idiag, lift, Dirichlet value and flag = 217 B; p_n once."""
        tot = 0.0
        for b, p in enumerate(self.per):
            nv, nt, ne = p["nv"], p["nt"], p["ne"]
            n2 = nv + ne
            iu = iters_per_step[b][0]
            tot += (2 + 2 * iu) * 64.0 * nt + 32.0 * n2 * iu + 217.0 * n2 + 8.0 * nv
        return tot

    def velocity_kernel_flops(self, iters_per_step) -> float:
        tot = 0.0
        for b, p in enumerate(self.per):
            iu = iters_per_step[b][0]
            tot += p["nt"] * ((1 + 2 * iu) * 600.0 + 1800.0) + (p["nv"] + p["ne"]) * iu * 60.0
        return tot

    def flops_per_step(self, iters_per_step) -> float:
        """fp64 operations (FMA = 2) of the matrix-free path per batch step (element operators only)."""
        tot = 0.0
        for b, p in enumerate(self.per):
            nt = p["nt"]
            iu, ip, im = iters_per_step[b]
            f_vel, f_mass, f_rhs1, f_rhs3 = 2 * 300.0, 2 * 60.0, 2 * 900.0, 2 * 80.0
            tot += nt * ((1 + 2 * iu) * f_vel + (1 + im) * f_mass + f_rhs1 + f_rhs3)
        return tot
