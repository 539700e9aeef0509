"""`VecEnv2DAirfoil` - B `Env2DAirfoil` environments stepped together on one GPU.

Same per-environment semantics as `meshdqn_amd.env.Env2DAirfoil` (= the reference's
Env2DAirfoil.py:318-428 step / reward / state logic), batched:

  host   mdq_remesh_host        vertex removal + Delaunay restoration + smooth(50)     (C++, threads over envs)
  host   mdq_env_topology_host  edges / P2 dofs / airfoil facets / removable / N-closest / state graph
  GPU    mdq_interpolate_snapshots   S snapshots onto every coarsened mesh              (one launch for all envs)
  GPU    mdq_probe_forces            2S force integrals per env                         (one launch)
  GPU    feature gather (torch indexing) and, optionally, the fused Q-network forward (mdq_gcn_forward)

All environments start from the same smoothed original mesh and share its ground truth and snapshots
(computed once by a base `Env2DAirfoil`, i.e. the reference's first `reset()`); terminated environments
are reset in place.  Triangulations are identical to the reference's as SETS of cells; the cell ORDER
(an artefact of Qhull in the reference) is the engine's own, so `edge_index` columns come in a different
order than in the single-environment class.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .env import Env2DAirfoil
from .mesh_ops import (DeviceTopologyBatch, HostTopologyBatch, remesh_batch, remesh_batch_gpu, remesh_workspace, smooth_batch_gpu,
                       smooth_env_gpu)


def _host_cores() -> int:
    """Usable host cores: the cgroup CPU quota when one is set (the MI355X boxes expose 256 logical CPUs
    under a 16-core quota), else the affinity mask.  The pool is sized 2x the quota: the mesh tasks are
    uneven and the extra workers hide the tail."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, 2 * int(q) // int(per)))
    except (OSError, ValueError):
        pass
    return n


class VecEnv2DAirfoil:
    FLOW_NRL = 4032         # row capacity of a chunk's list in the device-built tile maps (the kernels' LDS stage holds 4064 rows)

    def __init__(self, config, num_envs: int, compute_device="cuda", nthreads: int = 0, base_env: Env2DAirfoil | None = None,
                 auto_reset: bool = True, emax: int = 1536, flow_steps: int = 0, flow_rtol: float = 1e-10,
                 gpu_smoothing: bool = True, gpu_topology: bool = True, gpu_remesh: bool = True,
                 flow_overlap: bool = False, flow_pressure: str = "cg", flow_pcg_degree: int = 0):
        self.lib = _lib.load()
        self.B = int(num_envs)
        self.device = torch.device(compute_device)
        # workers of the host engine's persistent pool (one environment per task)
        self.nthreads = int(nthreads) if nthreads > 0 else max(1, min(_host_cores(), self.B))
        self.auto_reset = auto_reset
        self.gpu_smoothing = bool(gpu_smoothing)   # mdq_smooth (dataflow kernel) instead of the host loop
        # mdq_env_topology (device engine, bit-identical to the host engine): the host then only re-triangulates the
        # cavity of a removed vertex; needs the GPU smoothing path (coordinates stay on the device)
        self.gpu_topology = bool(gpu_topology) and self.gpu_smoothing
        # mdq_remesh (cavity re-triangulation + Lawson flips on the device): with it the meshes never leave the GPU;
        # the host arrays `coords / cells / nv / nt` are then read-only mirrors refreshed every step
        self.gpu_remesh = bool(gpu_remesh) and self.gpu_topology
        # S3 ("north-star step"): after every remesh, `flow_steps` IPCS steps on the coarsened mesh warm-started
        # from the interpolated last snapshot (0 = the reference's step, which never re-solves the flow)
        self.flow_steps, self.flow_rtol = int(flow_steps), float(flow_rtol)
        # S3 with the IPCS step of env step k running on a second stream BESIDE the Q-forward of step k and the vertex
        # removal / smoothing of step k + 1 (the state and the reward do not depend on the re-solved flow; the smoothing
        # kernel is one wave per mesh, the IPCS kernels take the other half of the chip): the flow stream works on a
        # private copy of the meshes and the drag / lift of the re-solved flow are reported one step later
        # (`infos["flow_lag"] = 1`).  The streams must sit on different hardware queues: GPU_MAX_HW_QUEUES >= 8 (set by
        # the package at import when the variable is not set)
        self.flow_overlap = bool(flow_overlap) and self.flow_steps > 0 and self.gpu_remesh
        # pressure solve of the S3 flow step on the freshly coarsened mesh: "cg" (Jacobi-CG, ~165 iterations, 0.28 ms on the
        # flow stream) or "direct" = what the reference does at a remesh (re-factorise: mdq_ipcs_factorize_pressure on the
        # device, 0.86 ms per batch, then a direct solve with 0 iterations; an environment whose mesh exceeds the kernel's
        # limits falls back to CG by itself).  Beside the smoothing kernel "direct" costs 1-5 % of the env step, 11 % of the
        # learning loop (its flow leg no longer hides next to the optimiser chain): one solve per mesh does not pay for a
        # factorisation, so "cg" is the default here; FlowSolver / deploy (thousands of steps per mesh) factorise
        if flow_pressure not in ("cg", "direct"):
            raise ValueError("flow_pressure: 'cg' or 'direct'")
        self.flow_pressure = flow_pressure
        self.flow_pcg_degree = int(flow_pcg_degree)     # Chebyshev degree of the pressure CG's preconditioner (0: Jacobi-CG kernel)
        base = base_env or Env2DAirfoil(config, compute_device=compute_device)
        self.base = base
        ap = config["agent_params"]
        self.N = int(ap["N_closest"])
        self.TIME_REWARD = float(ap["time_reward"])
        self.threshold = float(ap["threshold"])
        self.goal_vertices = float(ap["goal_vertices"])
        self.timesteps = int(ap["timesteps"])
        self.NEGATIVE_REWARD = -1.0
        self.gt_drag = np.asarray(base.gt_drag, dtype=np.float64)
        self.gt_lift = np.asarray(base.gt_lift, dtype=np.float64)
        self.S = len(base.original_u)
        topo0 = base._orig_topo
        self.x0 = topo0.coords.copy()
        self.cells0 = np.ascontiguousarray(topo0.cells, dtype=np.int32)
        self.NV, self.NT = topo0.nv, topo0.nt
        self.NE = topo0.ne
        self.NP = self.NV + self.NE
        self.EMAX = int(emax)
        self.polygon = np.ascontiguousarray(base.polygon, dtype=np.float64)
        self.interp = base._interp
        self.mu = base.flow_solver.mu
        B, NV, NT, NP, N = self.B, self.NV, self.NT, self.NP, self.N
        self.NAF = int(max((topo0.facet_tags() == 1).sum(), 1))
        # host state + outputs of the topology engine
        nbr_ptr = topo0.vertex_adjacency()[0]
        deg = np.zeros(64 * (NV // 64 + 1), np.int64)
        deg[:NV] = np.diff(nbr_ptr) + 1
        nse1 = int(64 * deg.reshape(-1, 64).max(axis=1).sum())        # SELL-64 entries of the P1 Laplacian, initial mesh
        nse1_cap = (int(1.2 * nse1) + 63) // 64 * 64
        self.topo = HostTopologyBatch(B, NV, NT, self.NE, self.NAF, N, self.EMAX, self.polygon,
                                      ipcs=self.flow_steps > 0 and not self.gpu_topology, nse1_cap=nse1_cap,
                                      nbo_cap=max(64, 2 * (2 * int((topo0.facet_tags() == 3).sum()) + 1)))
        self.dtopo = None
        # outflow rows of the facet term: two vertices + one edge dof per outflow facet, shared vertices counted once
        nbo_cap = max(64, 2 * (2 * int((topo0.facet_tags() == 3).sum()) + 1))
        if self.gpu_topology:
            self.dtopo = DeviceTopologyBatch(B, NV, NT, self.NE, self.NAF, N, self.EMAX, self.polygon, self.device,
                                             ipcs=self.flow_steps > 0 and not self.flow_overlap, nse1_cap=nse1_cap, nbo_cap=nbo_cap)
            if self.flow_overlap:
                # the flow stream's own engine: a private copy of the meshes, the full topology (with the IPCS index data,
                # which only the flow needs: 0.19 ms less on the critical path) and the IPCS step run there
                self._ftopo = DeviceTopologyBatch(B, NV, NT, self.NE, self.NAF, N, self.EMAX, self.polygon, self.device,
                                                  ipcs=True, nse1_cap=nse1_cap, flow_only=True, nbo_cap=nbo_cap)
        self._packed_host, self._packed_ev, self._pending, self._step_pending = None, torch.cuda.Event(), None, None
        self._restore_args = {}
        self._deferred_mirror = None
        self._node_ptr = torch.arange(B + 1, dtype=torch.int32, device=self.device) * N   # (constant: N rows per graph)
        # the initial mesh on the device: source rows of the in-place resets (mdq_restore_rows)
        self._x0_dev = torch.from_numpy(np.ascontiguousarray(topo0.coords, dtype=np.float64)).to(self.device)
        self._cells0_dev = torch.from_numpy(np.ascontiguousarray(topo0.cells, dtype=np.int32)).to(self.device)
        self._nv0_dev = torch.tensor([NV], dtype=torch.int32, device=self.device)
        self._nt0_dev = torch.tensor([NT], dtype=torch.int32, device=self.device)
        self._zero_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        if self.gpu_remesh:
            self._rstat = torch.zeros(B, dtype=torch.int32, device=self.device)
            self._rstat_host = torch.zeros(B, dtype=torch.int32, pin_memory=True)
            self._mirror_stream = torch.cuda.Stream(device=self.device)
            self._mirror_ev = torch.cuda.Event()
            self._mirror_done = torch.cuda.Event()
        if self.flow_steps > 0:
            self._init_flow(base)
        self.coords, self.cells, self.nv, self.nt, self.offset = (self.topo.coords, self.topo.cells, self.topo.nv,
                                                                   self.topo.nt, self.topo.offset)
        self.h = self.topo.h   # (device engine: only nsel / n_closest / coord_map / nedges / ne are mirrored here)
        self.steps = np.zeros(B, np.int64)
        self.initial_num_node = NV
        self.new_drags = np.zeros((B, self.S))
        self.new_lifts = np.zeros((B, self.S))
        self.reset_all()

    # ------------------------------------------------------------------
    def _init_flow(self, base):
        """Device arrays + descriptor(s) of the matrix-free IPCS path (mode 3, CG pressure) over the batch."""
        dev, tp = self.device, (self.dtopo if self.gpu_topology else self.topo)
        if self.flow_overlap:
            tp = self._ftopo
        B, NV, NT, NE, NP = self.B, self.NV, self.NT, self.NE, self.NP

        def z(*shape):
            return torch.zeros(shape, dtype=torch.float64, device=dev)

        own = dict(geom=z(B, 5, NT), bo_val=z(B, tp.NBE, 4), lift1=z(B, NP, 2), lift3=z(B, NP, 2), idiag1=z(B, NP, 2),
                   sdiagM=z(B, NP), sdiagK=z(B, NV), K1s=z(B, tp.NSE1), u_n=z(B, NP, 2), p_n=z(B, NV))
        nwork = int(self.lib.mdq_ipcs_workspace_doubles(B, NV, NT, NE))
        own["work"] = z(nwork)
        fs = base.flow_solver
        if self.gpu_topology:   # the device engine's outputs ARE the descriptor's index arrays
            index_sets = [dict(tp.ti)]
        else:
            index_sets = [{k: torch.from_numpy(a).to(dev) for k, a in tp.hi.items()}]
        self.flow_descs, self.flow_ts = [], []
        for idx in index_sets:
            t = dict(idx)
            t.update(own)
            d = _lib.IpcsDesc()
            d.B, d.NV, d.NT, d.NE, d.N2, d.NAF = B, NV, NT, NE, NP, self.NAF
            d.NSE1, d.NBO, d.NBE = tp.NSE1, tp.NBO, tp.NBE
            d.mu, d.rho, d.dt, d.rtol = fs.mu, fs.rho, fs.dt_value, self.flow_rtol
            d.maxit_u, d.maxit_p, d.maxit_m = 200, 4000, 200
            # mode 3 (LDS-resident vectors) on the lab meshes; a mesh beyond its limits (the red-refined ones): auto, i.e. the
            # element tiles with global vectors (mode 5, here without tile maps: the dof <- slot lists of the topology engine)
            d.mode, d.pd_enabled = (3 if NP <= 3584 else -1), 0
            d.pcg_degree = int(getattr(self, "flow_pcg_degree", 0))
            for name, _typ in _lib.IpcsDesc._fields_:
                if name in t:
                    setattr(d, name, t[name].data_ptr())
            d.work_doubles = nwork
            # meshes beyond the LDS-resident modes (auto: the element tiles, modes 5 / 7): the tile maps of every coarsened mesh are
            # built on the device in front of the IPCS step (mdq_ipcs_build_tile_maps: row lists + packed local maps) - without
            # them the element results of every operator application go through 0.6 MB of global scratch per environment
            # (MDQ_NO_DEVICE_TILE_MAPS=1: that path, A / B switch)
            if NP > 3584 and self.gpu_topology and os.environ.get("MDQ_NO_DEVICE_TILE_MAPS", "") != "1":
                nch = (NT + 1023) // 1024
                nrl = min(self.FLOW_NRL, NP)
                if (nch + 4) * ((NP + 15) & ~15) + 256 <= 160 * 1024:
                    t["mf_rlist"] = torch.zeros((B, nch, nrl, 2), dtype=torch.int32, device=dev)
                    t["mf_rcnt"] = torch.zeros((B, nch), dtype=torch.int32, device=dev)
                    t["mf_lpos"] = torch.zeros((B, 6, NT), dtype=torch.int32, device=dev)
                    for k in ("mf_rlist", "mf_rcnt", "mf_lpos"):
                        setattr(d, k, t[k].data_ptr())
                    d.NRL, d.rl_flags = nrl, 1
                    d.mf_scat = d.mf_tptr = None
                    self._flow_tile_maps = True
            self.flow_descs.append(d)
            self.flow_ts.append(t)
        self.flow_t, self.flow_desc = self.flow_ts[0], self.flow_descs[0]
        if self.flow_pressure == "direct":      # outputs of mdq_ipcs_factorize_pressure (capacities: its limits)
            from .ipcs_batch import IpcsBatch
            cap = IpcsBatch.PD_DEVICE_CAP
            i32, f64 = torch.int32, torch.float64
            zz = lambda n, dt_: torch.zeros((B, n), dtype=dt_, device=dev)   # noqa: E731
            pdt = dict(pd_hdr=zz(4, i32), pd_node=zz(NV, i32), pd_meta=zz(cap["NPART"] * 6, i32), pd_rowblk=zz(NV, i32),
                       pd_W=zz(cap["NPW"], f64), pd_F=zz(cap["NPF"], f64), pd_gidx=zz(cap["NPGI"], i32),
                       pd_Sinv=zz(cap["NPS"], f64), pd_gk_ptr=zz(NV + 1, i32), pd_gk_col=zz(cap["NPGK"], i32),
                       pd_gk_val=zz(cap["NPGK"], f64))
            for t, d in zip(self.flow_ts, self.flow_descs):
                t.update(pdt)
                for k, a in pdt.items():
                    setattr(d, k, a.data_ptr())
                d.NPART, d.NPW, d.NPF, d.NPGI, d.NPS, d.NPGK = (cap[k] for k in ("NPART", "NPW", "NPF", "NPGI", "NPS", "NPGK"))
                d.pd_enabled = 1
            self.flow_pd_status = torch.zeros(B, dtype=i32, device=dev)
        self.flow_iters = torch.zeros((B, 3), dtype=torch.int32, device=dev)
        # sticky status words of the flow legs (mdq_ipcs_desc.status, ABI 7): a leg that gave up - team barrier time-out of the
        # two-workgroup operator modes - is an ERROR at the next read-back (rollout_end / flow_wait), not a silent NaN
        self.flow_status = torch.zeros(B, dtype=torch.int32, device=dev)
        for d in self.flow_descs:
            d.status = self.flow_status.data_ptr()
        self.flow_drag = np.zeros((B, self.flow_steps))
        self.flow_lift = np.zeros((B, self.flow_steps))
        if self.flow_overlap:
            from .streams import role_streams
            self._flow_stream = role_streams(dev)["flow"]   # (the process's flow stream: fixed creation order, one probe)
            self._flow_ready = torch.cuda.Event()
            self._late_handover = os.environ.get("MDQ_LATE_HANDOVER", "") == "1"      # (A / B switch of the early mesh hand-over)
            # the last IPCS kernel writes drag / lift of the leg straight into the page-locked result buffers (2 KB over the bus)
            # instead of two device-to-host copies behind it (12 us of the flow chain); MDQ_FLOW_RESULT_COPY=1: the copies
            self._flow_direct_results = os.environ.get("MDQ_FLOW_RESULT_COPY", "") != "1"
            self._handover_in_kernel = os.environ.get("MDQ_HANDOVER_COPY", "") != "1"    # (A / B: the copy launch of the first version)
            # page-locked result buffers (two: the results of step k are read while step k + 1 is in flight) + events
            self._flow_res = [dict(host=torch.zeros((2, B, self.flow_steps), dtype=torch.float64, pin_memory=True),
                                   done=torch.cuda.Event()) for _ in range(2)]
            self._flow_n = 0          # flows launched
            self._flow_prev = None
            # TWO sets of the flow engine's inputs (private meshes + start fields): the main stream fills set k % 2 for
            # flow k while flow k - 1 still reads the other one.  With one set the hand-over was a serial chain - flow
            # k - 1 ends -> main copies -> flow k starts, two event round trips + the copy per step on the flow's
            # critical path - and the S3 step lasted (flow leg + ~70 us) instead of max(main chain, flow leg).
            ft, t0 = self._ftopo, self.flow_ts[0]
            self._flow_in = [dict(coords=ft.coords, cells=ft.cells, nv=ft.nv, nt=ft.nt, u_n=t0["u_n"], p_n=t0["p_n"],
                                  # the main engine's cell dofs / edge counts of the same meshes: the flow's topology run takes
                                  # its edge numbering from them (no second hash pass)
                                  cell_dofs=torch.zeros_like(self.dtopo.t["cell_dofs"]), ne=torch.zeros_like(self.dtopo.t["ne"]))]
            self._flow_in.append({k: torch.zeros_like(a) for k, a in self._flow_in[0].items()})

    def _flow(self, keep, out_u, out_p):
        """`flow_steps` IPCS steps on every (coarsened) mesh, warm-started from the interpolated last snapshot."""
        t, d = self.flow_ts[0], self.flow_descs[0]
        if not self.gpu_topology:
            for kk in self.topo.hi:
                t[kk].copy_(self.topo.pinned[kk], non_blocking=True)
        if self.flow_overlap:
            return self._flow_overlapped(out_u, out_p)
        for kk in ("coords", "cell_dofs", "af_facets", "nv", "nt", "ne", "naf"):
            setattr(d, kk, keep[kk].data_ptr())
        return self._flow_launch(t, d, keep, out_u, out_p)

    def _flow_reset(self, d):
        # no initial-guess history on a new mesh: the counters inside the workspace (not a fill of its 100 MB) + the
        # iteration counters, one small launch
        _lib.check(self.lib.mdq_ipcs_reset_history(C.byref(d), self.flow_iters.data_ptr(), _lib.stream_ptr()),
                   "mdq_ipcs_reset_history")

    def _flow_launch(self, t, d, keep, out_u, out_p, before_evolve=None, reset=True, out=None):
        """`reset=False`: the caller has already enqueued `_flow_reset` (ahead of a wait: off the leg's critical path).
        `out` = (drag, lift) tensors the kernels write - page-locked host tensors are written over the bus directly, which
        saves the two result copies at the end of the leg."""
        if out_u is not None:
            t["u_n"].copy_(out_u[:, self.S - 1])
            t["p_n"].copy_(out_p[:, self.S - 1])
        if reset:
            self._flow_reset(d)
        _lib.check(self.lib.mdq_ipcs_setup_matfree(C.byref(d), _lib.stream_ptr()), "mdq_ipcs_setup_matfree")
        if getattr(self, "_flow_tile_maps", False):
            _lib.check(self.lib.mdq_ipcs_build_tile_maps(C.byref(d), None, _lib.stream_ptr()), "mdq_ipcs_build_tile_maps")
        if self.flow_pressure == "direct":
            _lib.check(self.lib.mdq_ipcs_factorize_pressure(C.byref(d), self.flow_pd_status.data_ptr(), _lib.stream_ptr()),
                       "mdq_ipcs_factorize_pressure")
        if out is not None:
            drag, lift = out
        else:
            drag = torch.empty((self.B, self.flow_steps), dtype=torch.float64, device=self.device)
            lift = torch.empty_like(drag)
        if before_evolve is not None:       # (the set-up above reads the mesh only; the warm start is needed from here on)
            before_evolve()
        _lib.check(self.lib.mdq_ipcs_evolve(C.byref(d), self.flow_steps, drag.data_ptr(), lift.data_ptr(),
                                            self.flow_iters.data_ptr(), _lib.stream_ptr()), "mdq_ipcs_evolve")
        self._flow_keep = keep       # device buffers the descriptor points at
        return drag, lift

    def _flow_handover(self, out_u, out_p):
        """First half of the overlapped flow leg: the input set the flow stream will read (filled alternately: flow k - 1
        may still be reading the other one) and the (destination, source) pairs of the hand-over - the meshes, the main
        engine's edge numbering and the warm start (last interpolated snapshot), copied BEFORE an in-place reset of a
        terminated environment rewrites them."""
        dt = self.dtopo
        main = torch.cuda.current_stream(self.device)
        fin = self._flow_in[self._flow_n % 2]
        if self._flow_n >= 2 and self._flow_prev is not None:          # flow k - 2 read this input set (long finished)
            main.wait_event(self._flow_res[self._flow_n % 2]["done"])
        su, sp_ = out_u[:, self.S - 1], out_p[:, self.S - 1]
        pairs = [(fin["coords"], dt.coords), (fin["cells"], dt.cells), (fin["nv"], dt.nv), (fin["nt"], dt.nt), (fin["u_n"], su), (fin["p_n"], sp_),
                 (fin["cell_dofs"], dt.t["cell_dofs"]), (fin["ne"], dt.t["ne"])]
        return fin, pairs

    def _flow_handover_target(self):
        """Before the main topology run of a device-resident step: the flow input set this step fills becomes the second
        output set of the topology kernel (meshes + edge numbering written by that launch itself: no copy launch on the main
        chain, 6 us + two event gaps per step)."""
        main = torch.cuda.current_stream(self.device)
        fin = self._flow_in[self._flow_n % 2]
        if self._flow_n >= 2 and self._flow_prev is not None:          # flow k - 2 read this input set (long finished)
            main.wait_event(self._flow_res[self._flow_n % 2]["done"])
        self.dtopo.set_handover(fin["coords"], fin["cells"], fin["nv"], fin["nt"], fin["cell_dofs"], fin["ne"])

    def _flow_handover_mesh(self):
        """Device-resident step, EARLY half of the hand-over: meshes + the main engine's edge numbering are copied (one
        launch on the main stream) right after the main topology run, so that the flow stream can derive its own topology
        and the operator set-up while the main stream still interpolates / evaluates the step - handing everything over
        inside `mdq_env_finish` left the flow stream idle for ~60 us of every step (profiles/r04_timeline_s3_step.txt).
        The warm start (u / p windows) follows in `mdq_env_finish` as before."""
        dt = self.dtopo
        main = torch.cuda.current_stream(self.device)
        fin = self._flow_in[self._flow_n % 2]
        if not self._handover_in_kernel and self._flow_n >= 2 and self._flow_prev is not None:   # flow k - 2 read this input set
            main.wait_event(self._flow_res[self._flow_n % 2]["done"])
        if self._handover_in_kernel:
            # the main topology kernel has written this input set itself (`_flow_handover_target`, mdq_topo_handover)
            dt.set_handover()                    # (this launch only: a later host-driven step() must not write there)
            if getattr(self, "_flow_mesh_ready", None) is None:
                self._flow_mesh_ready = torch.cuda.Event()
            self._flow_mesh_ready.record(main)
            self._flow_fin_early = fin
            return
        pairs = [(fin["coords"], dt.coords), (fin["cells"], dt.cells), (fin["nv"], dt.nv), (fin["nt"], dt.nt),
                 (fin["cell_dofs"], dt.t["cell_dofs"]), (fin["ne"], dt.t["ne"])]
        n = len(pairs)
        vp, i64 = C.c_void_p * n, C.c_int64 * n
        nb = [src.numel() * src.element_size() for _, src in pairs]
        for dst, src in pairs:
            if dst.shape != src.shape or dst.dtype != src.dtype or not dst.is_contiguous() or not src.is_contiguous():
                raise ValueError("flow hand-over: buffers of different shapes")
        _lib.check(self.lib.mdq_copy_strided(n, vp(*[d_.data_ptr() for d_, _ in pairs]), vp(*[s_.data_ptr() for _, s_ in pairs]),
                                             i64(*([1] * n)), i64(*nb), i64(*nb), i64(*nb), _lib.stream_ptr()), "mdq_copy_strided")
        if getattr(self, "_flow_mesh_ready", None) is None:
            self._flow_mesh_ready = torch.cuda.Event()
        self._flow_mesh_ready.record(main)
        self._flow_fin_early = fin

    def _flow_start(self, fin, mesh_early=False):
        """Second half: the hand-over is enqueued on the main stream - topology (edges from the main engine), matrix-free
        set-up and the IPCS step(s) follow on the flow stream; results land in page-locked memory.  `mesh_early`: the
        meshes were handed over by `_flow_handover_mesh` - topology and set-up wait for THAT, only the IPCS step for the
        rest."""
        ft = self._ftopo
        t, d = self.flow_ts[0], self.flow_descs[0]
        main = torch.cuda.current_stream(self.device)
        self._flow_ready.record(main)
        keep = dict(coords=fin["coords"], cell_dofs=ft.t["cell_dofs"], af_facets=ft.t["af_facets"], nv=fin["nv"], nt=fin["nt"],
                    ne=ft.t["ne"], naf=ft.t["naf"], u_n=fin["u_n"], p_n=fin["p_n"])
        for kk, v in keep.items():
            setattr(d, kk, v.data_ptr())
        for kk in ("coords", "cells", "nv", "nt"):            # the flow's topology engine reads the same set
            setattr(ft.desc, kk, fin[kk].data_ptr())
        ft.take_edges_from(fin["cell_dofs"], fin["ne"])
        t["u_n"], t["p_n"] = fin["u_n"], fin["p_n"]
        res = self._flow_res[self._flow_n % 2]
        direct = self._flow_direct_results and tuple(res["host"][0].shape) == (self.B, self.flow_steps)
        with torch.cuda.stream(self._flow_stream):
            self._flow_reset(d)                  # (behind flow k - 1, in FRONT of the wait for this step's meshes)
            self._flow_stream.wait_event(self._flow_mesh_ready if mesh_early else self._flow_ready)
            fe = getattr(self, "flow_events", None)     # (tools: HIP events around the leg, on the flow stream)
            if fe is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if getattr(self, "_flow_tile_maps", False):
                # the private copy's cells in a spatial order (chunks of 1 024 triangles that share their rows: what the tile maps
                # of the IPCS step need), the main engine's cell dofs of the same cells permuted alike
                _lib.check(self.lib.mdq_flow_sort_cells(self.B, self.NV, self.NT, fin["coords"].data_ptr(), fin["nv"].data_ptr(),
                                                        fin["nt"].data_ptr(), fin["cells"].data_ptr(), fin["cell_dofs"].data_ptr(),
                                                        _lib.stream_ptr()), "mdq_flow_sort_cells")
            ft.run(check=False)                  # (same meshes, same deterministic kernel as the main stream's run)
            drag, lift = self._flow_launch(t, d, keep, None, None,
                                           (lambda: self._flow_stream.wait_event(self._flow_ready)) if mesh_early else None,
                                           reset=False, out=(res["host"][0], res["host"][1]) if direct else None)
            if fe is not None:
                e1.record()
                fe.append((e0, e1))
            if not direct:
                res["host"][0].copy_(drag, non_blocking=True)
                res["host"][1].copy_(lift, non_blocking=True)
            res["done"].record(self._flow_stream)
        self._flow_prev = self._flow_n % 2
        self._flow_n += 1

    def _flow_overlapped(self, out_u, out_p):
        """The IPCS leg on the flow stream: the meshes are copied to the flow's own engine (on the main stream, behind the
        last flow), which derives topology + IPCS index data itself (host-driven `step()`; the device-resident step hands
        the rows over inside `mdq_env_finish`)."""
        fin, pairs = self._flow_handover(out_u, out_p)
        # meshes + the warm start in ONE launch (six torch copies were ~50 us of the main chain)
        n = len(pairs)
        vp, i64 = C.c_void_p * n, C.c_int64 * n
        rows, rb, ss, ds = [], [], [], []
        for dst, src in pairs:
            if dst.shape != src.shape or dst.dtype != src.dtype or not dst.is_contiguous():
                raise ValueError("flow hand-over: buffers of different shapes")
            if src.is_contiguous():
                rows.append(1); rb.append(src.numel() * src.element_size()); ss.append(rb[-1]); ds.append(rb[-1])
            else:                                   # a snapshot slice: contiguous per environment
                per = src[0].numel() * src.element_size()
                if not src[0].is_contiguous():
                    raise ValueError("flow hand-over: rows of the source must be contiguous")
                rows.append(src.shape[0]); rb.append(per); ss.append(src.stride(0) * src.element_size()); ds.append(per)
        _lib.check(self.lib.mdq_copy_strided(n, vp(*[d_.data_ptr() for d_, _ in pairs]), vp(*[s_.data_ptr() for _, s_ in pairs]),
                                             i64(*rows), i64(*rb), i64(*ss), i64(*ds), _lib.stream_ptr()), "mdq_copy_strided")
        self._flow_start(fin)
        return None, None

    def flow_wait(self):
        """Overlap mode: wait for the IPCS step launched last and return its (drag, lift), or None."""
        if not self.flow_overlap or self._flow_prev is None:
            return None
        res = self._flow_res[self._flow_prev]
        res["done"].synchronize()
        fd, fl = res["host"][0].numpy().copy(), res["host"][1].numpy().copy()
        _check_flow_forces(fd, fl, "flow_wait", self.flow_status.cpu().numpy())
        return fd, fl

    # ------------------------------------------------------------------
    def _reset_env(self, b):
        self.coords[b] = self.x0
        self.cells[b] = self.cells0
        self.nv[b], self.nt[b] = self.NV, self.NT
        self.offset[b] = 0
        self.steps[b] = 0

    def reset_all(self):
        if getattr(self, "flow_overlap", False) and getattr(self, "_flow_prev", None) is not None:
            # results of a flow leg launched before the reset belong to meshes that no longer exist: the first step
            # after a reset (or a stream calibration) reports no "previous step" forces
            self.flow_wait()
            self._flow_prev = None
            self._flow_prev2 = None
        for b in range(self.B):
            self._reset_env(b)
        if self.gpu_topology:
            self._upload_mesh()
        self._refresh()
        # every environment restarts from the same mesh: cache its derived data (row 0) for in-place resets
        self._init_cache = dict(h={k: a[0].copy() for k, a in self.h.items()}, u=self.u[0].clone(), p=self.p[0].clone(),
                                drags=self.new_drags[0].copy(), lifts=self.new_lifts[0].copy())
        if self.gpu_topology:
            self._init_cache["dev"] = {k: self.dtopo.t[k][0].clone() for k in self._STATE_KEYS}
        return self.get_state()

    _STATE_KEYS = ("n_closest", "nsel", "coord_map", "nedges", "edge_src", "edge_dst")   # what get_state reads

    def _upload_mesh(self):
        """Host mesh (page-locked) -> the device engine's input tensors, asynchronously on the current stream."""
        dt, pin = self.dtopo, self.topo.pinned
        dt.coords.copy_(pin["coords"], non_blocking=True)
        dt.cells.copy_(pin["cells"], non_blocking=True)
        dt.nv.copy_(pin["nv"], non_blocking=True)
        dt.nt.copy_(pin["nt"], non_blocking=True)
        dt.offset.copy_(torch.from_numpy(self.offset), non_blocking=False)

    def _restore_initial(self, idx):
        """Reset environments `idx` in place from the cached initial-mesh data (no recomputation)."""
        c = self._init_cache
        idx = np.asarray(idx)
        self._deferred_mirror = idx     # coords / cells host mirrors: written after the next state has been launched
        self.nv[idx], self.nt[idx] = self.NV, self.NT
        self.offset[idx] = 0
        self.steps[idx] = 0
        # (device engine: only these are mirrored on the host, see _refresh_collect; the others are dead rows)
        keys = ("nsel", "n_closest", "coord_map", "nedges", "ne") if self.gpu_topology else tuple(self.h)
        for k in keys:
            self.h[k][idx] = c["h"][k]
        self.new_drags[idx] = c["drags"]
        self.new_lifts[idx] = c["lifts"]
        # device side: ONE launch restores the rows of every tensor (a dozen index_put launches otherwise).  The
        # argument arrays are built once; only the two per-step buffers (u, p) change their addresses.
        ra = self._restore_arg_arrays()
        ti = torch.from_numpy(idx.astype(np.int32)).to(self.device)
        _lib.check(self.lib.mdq_restore_rows(ra["n"], ra["dst"], ra["src"], ra["nbytes"], int(ti.numel()), ti.data_ptr(),
                                             _lib.stream_ptr()), "mdq_restore_rows")

    def _restore_arg_arrays(self):
        c = self._init_cache
        ra = self._restore_args.get(0)
        if ra is None or ra["coords"] != self._coords_dev.data_ptr():
            pairs = [(self.u, c["u"]), (self.p, c["p"]), (self._coords_dev, self._x0_dev)]
            if self.gpu_topology:
                pairs += [(self.dtopo.t[k], c["dev"][k]) for k in self._STATE_KEYS]
            if self.gpu_remesh:     # the device holds the meshes: reset them there as well
                dt = self.dtopo
                pairs += [(dt.cells, self._cells0_dev), (dt.nv, self._nv0_dev), (dt.nt, self._nt0_dev),
                          (dt.offset, self._zero_dev)]
            n = len(pairs)
            nbytes = [a[0].numel() * a.element_size() for a, _ in pairs]
            for (a, b_), nb in zip(pairs, nbytes):
                assert a.is_contiguous() and b_.is_contiguous() and b_.numel() * b_.element_size() == nb and a.dtype == b_.dtype
            ra = self._restore_args[0] = dict(n=n, dst=(C.c_void_p * n)(*[a.data_ptr() for a, _ in pairs]),
                                              src=(C.c_void_p * n)(*[b_.data_ptr() for _, b_ in pairs]),
                                              nbytes=(C.c_int64 * n)(*nbytes), coords=self._coords_dev.data_ptr(),
                                              keep=pairs)
        ra["dst"][0], ra["dst"][1] = self.u.data_ptr(), self.p.data_ptr()
        return ra

    def _refresh(self):
        """Topology + selection, snapshot interpolation + forces on the GPU, for all envs."""
        self._refresh_launch()
        self._refresh_collect()

    def _refresh_launch(self, readback=True, defer_flow=False, after_topology=None, sparse=0, before_topology=None):
        """Everything of `_refresh` that is enqueued on the stream, up to the asynchronous read-back (`readback=False`:
        the device-resident rollout keeps the results on the device)."""
        dev, h = self.device, self.h
        B, NV, NT, NP = self.B, self.NV, self.NT, self.NP
        if self.gpu_topology:
            dt = self.dtopo
            if before_topology is not None:
                before_topology()
            try:
                dt.run(check=False)                 # status is read back with the other results below
            except Exception:
                dt.set_handover()                   # (a hand-over armed by `before_topology` must not outlive a failed launch:
                raise                               #  a later host-driven step() would overwrite the flow leg's input set)
            if after_topology is not None:
                after_topology()
            t_pts, np1 = dt.t["points"], dt.nv
            npts, npts_extra = np1, dt.t["ne"]          # (P2 points = vertices + edges: added inside the kernel)
        else:
            torch.cuda.current_stream(dev).synchronize()   # pending async uploads read the arrays the engine rewrites
            self.topo.run(self.nthreads)
            up = self.topo.upload
            t_pts = up("points", dev)
            np1 = up("nv", dev)
            npts, npts_extra = np1 + up("ne", dev), None
        it = self.interp
        # two persistent (ping-pong) result sets, zero-filled once: the interpolation writes every valid dof of the current
        # meshes and nothing reads the padding behind them (INVARIANT: rows behind nv / np2 of a set hold stale values of
        # earlier, larger meshes, not zeros) (two 45 MB fills per step were ~30 us of the main chain); the
        # set of the previous step stays intact for whoever still holds it
        if getattr(self, "_interp_bufs", None) is None:
            self._interp_bufs = [(torch.zeros((B, self.S, NP, 2), dtype=torch.float64, device=dev),
                                  torch.zeros((B, self.S, NV), dtype=torch.float64, device=dev)) for _ in range(2)]
            self._interp_i = 0
        self._interp_i ^= 1
        out_u, out_p = self._interp_bufs[self._interp_i]
        d = _lib.InterpDesc()
        d.B, d.S, d.NP, d.NP1 = B, self.S, NP, NV
        d.src_nv, d.src_nt, d.src_n2 = it.topo.nv, it.topo.nt, it.topo.np2
        d.gnx, d.gny, d.x0, d.y0, d.inv_hx, d.inv_hy = it.grid
        d.npts, d.np1, d.points = npts.data_ptr(), np1.data_ptr(), t_pts.data_ptr()
        d.npts_extra = None if npts_extra is None else npts_extra.data_ptr()
        for k, v in it.t.items():
            setattr(d, k, v.data_ptr())
        d.out_u, d.out_p, d.out_cell = out_u.data_ptr(), out_p.data_ptr(), None
        if sparse and self.gpu_topology:
            # device-resident step: only the entries it reads (vertices, the last snapshot's edge values for the flow leg's warm
            # start, the edges of the airfoil-facet cells for the force integrals) - the edge values of the other snapshots, 60 %
            # of the kernel's work, are read by nothing; the host-driven step() keeps the full fields (self.u is public there)
            d.sparse, d.NT, d.NAF = int(sparse), NT, self.NAF
            d.af_facets, d.naf, d.cell_dofs = dt.t["af_facets"].data_ptr(), dt.t["naf"].data_ptr(), dt.t["cell_dofs"].data_ptr()
        _lib.check(self.lib.mdq_interpolate_snapshots(C.byref(d), _lib.stream_ptr()), "mdq_interpolate_snapshots")
        self._interp_last = (d, int(d.sparse), (out_u, out_p, t_pts, npts, np1, npts_extra))     # (rollout_end completes a sparse field)
        # forces: light mesh descriptor over the batch
        md = _lib.IpcsDesc()
        md.B, md.NV, md.NT, md.NE, md.N2, md.NAF = B, NV, NT, self.NE, NP, self.NAF
        md.mu = self.mu
        if self.gpu_topology:
            t_coords = dt.coords
            keep = dict(coords=t_coords, cell_dofs=dt.t["cell_dofs"], af_facets=dt.t["af_facets"], nv=np1, nt=dt.nt,
                        ne=dt.t["ne"], naf=dt.t["naf"])
        else:
            t_coords = up("coords", dev)
            keep = dict(coords=t_coords, cell_dofs=up("cell_dofs", dev), af_facets=up("af_facets", dev), nv=np1,
                        nt=up("nt", dev), ne=up("ne", dev), naf=up("naf", dev))
        for k, v in keep.items():
            setattr(md, k, v.data_ptr())
        drag = torch.empty((B, self.S), dtype=torch.float64, device=dev)
        lift = torch.empty_like(drag)
        _lib.check(self.lib.mdq_probe_forces(C.byref(md), self.S, out_u.data_ptr(), out_p.data_ptr(), drag.data_ptr(),
                                             lift.data_ptr(), _lib.stream_ptr()), "mdq_probe_forces")
        self.u, self.p, self._coords_dev = out_u, out_p, t_coords
        self._dev_drag, self._dev_lift = drag, lift
        fd = fl = None
        if self.flow_steps > 0 and not (defer_flow and self.flow_overlap):   # (deferred: handed over by mdq_env_finish)
            fd, fl = self._flow(keep, out_u, out_p)
            if not self.gpu_topology:
                self.flow_drag, self.flow_lift = fd.cpu().numpy(), fl.cpu().numpy()
        if not readback:
            self._pending = None
            return
        if self.gpu_topology:
            # one read-back for everything the host logic needs: forces + status + the small integer mirrors
            N = self.N
            parts = [drag.reshape(-1).view(torch.int32), lift.reshape(-1).view(torch.int32)]
            nflow = 0
            if fd is not None:      # forces of the re-solved flow ride along (no extra synchronisation)
                parts += [fd.reshape(-1).view(torch.int32), fl.reshape(-1).view(torch.int32)]
                nflow = fd.numel()
            nf = 2 * B * self.S + 2 * nflow
            packed = torch.cat(parts + [dt.status, dt.t["nsel"], dt.t["nedges"], dt.t["ne"], dt.t["coord_map"].reshape(-1),
                                        dt.t["n_closest"].reshape(-1)])
            if self._packed_host is None or self._packed_host.numel() != packed.numel():
                self._packed_host = torch.empty(packed.numel(), dtype=torch.int32, pin_memory=True)
            self._packed_host.copy_(packed, non_blocking=True)      # page-locked: the copy is queued behind the kernels
            self._packed_ev.record(torch.cuda.current_stream(dev))
            self._pending = (nf, nflow, None if fd is None else tuple(fd.shape))
        else:
            self._pending = (drag, lift)

    def _refresh_collect(self):
        """Wait for the read-back of `_refresh_launch` and update the host mirrors."""
        h, B, N = self.h, self.B, self.N
        if self.gpu_topology:
            nf, nflow, fshape = self._pending
            self._packed_ev.synchronize()
            packed = self._packed_host.numpy()
            fl64 = packed[:2 * nf].view(np.float64)
            ints = packed[2 * nf:]
            if fshape is not None:
                self.flow_drag = fl64[2 * B * self.S:2 * B * self.S + nflow].reshape(fshape).copy()
                self.flow_lift = fl64[2 * B * self.S + nflow:].reshape(fshape).copy()
            self.new_drags = fl64[:B * self.S].reshape(B, self.S).copy()
            self.new_lifts = fl64[B * self.S:2 * B * self.S].reshape(B, self.S).copy()
            st = ints[:B]
            if (st != 0).any():
                raise _lib.MeshDQNHipError(f"topology kernel failed: env {np.flatnonzero(st)} status {st[st != 0]}")
            h["nsel"][...] = ints[B:2 * B]
            h["nedges"][...] = ints[2 * B:3 * B]
            h["ne"][...] = ints[3 * B:4 * B]
            h["coord_map"][...] = ints[4 * B:4 * B + B * N].reshape(B, N)
            h["n_closest"][...] = ints[4 * B + B * N:].reshape(B, N)
        else:
            drag, lift = self._pending
            self.new_drags = drag.cpu().numpy().copy()
            self.new_lifts = lift.cpu().numpy().copy()
        self._pending = None

    # ------------------------------------------------------------------
    def get_state(self):
        """dict of device tensors: x (B,N,2+3S) f32, esrc/edst (sumE,) i32 local ids, edge_ptr (B+1,) i32,
        node_ptr (B+1,) i32 - directly consumable by the fused Q-network forward - plus host copies of
        n_closest / coord_map / nedges."""
        dev, h, B, N, S = self.device, self.h, self.B, self.N, self.S
        x = torch.empty((B, N, 2 + 3 * S), dtype=torch.float32, device=dev)
        if self.gpu_topology:
            nc, nsel = self.dtopo.t["n_closest"], self.dtopo.t["nsel"]
        else:
            nc, nsel = self.topo.upload("n_closest", dev), self.topo.upload("nsel", dev)
        _lib.check(self.lib.mdq_state_features(B, N, S, self.NV, self.NP, self._coords_dev.data_ptr(), self.u.data_ptr(),
                                               self.p.data_ptr(), nc.data_ptr(), nsel.data_ptr(), x.data_ptr(),
                                               _lib.stream_ptr()), "mdq_state_features")
        ne = h["nedges"].astype(np.int64)
        edge_ptr = np.zeros(B + 1, np.int32)
        edge_ptr[1:] = np.cumsum(ne)
        if self.gpu_topology:
            # packed edge lists from the padded (B,EMAX) device arrays: one small kernel driven by the offsets (the counts
            # are on the host already, so the output size is known without a device-to-host synchronisation)
            total = int(edge_ptr[-1])
            edge_ptr_d = torch.from_numpy(edge_ptr).to(dev)
            esrc_d = torch.empty(total, dtype=torch.int32, device=dev)
            edst_d = torch.empty(total, dtype=torch.int32, device=dev)
            _lib.check(self.lib.mdq_compact_edges(B, self.EMAX, self.dtopo.t["edge_src"].data_ptr(),
                                                  self.dtopo.t["edge_dst"].data_ptr(), edge_ptr_d.data_ptr(),
                                                  esrc_d.data_ptr(), edst_d.data_ptr(), _lib.stream_ptr()), "mdq_compact_edges")
        else:
            live = np.arange(self.EMAX)[None, :] < ne[:, None]      # (B,EMAX) valid edge slots, row-major = env order
            esrc_d, edst_d = torch.from_numpy(h["edge_src"][live]).to(dev), torch.from_numpy(h["edge_dst"][live]).to(dev)
        pad = {}
        if self.gpu_topology:   # the padded (B,EMAX) edge lists as well (views of the engine's output, valid until
            pad = dict(edge_src_pad=self.dtopo.t["edge_src"], edge_dst_pad=self.dtopo.t["edge_dst"])   # the next step)
        return dict(x=x, esrc=esrc_d, edst=edst_d, **pad,
                    edge_ptr=edge_ptr_d if self.gpu_topology else torch.from_numpy(edge_ptr).to(dev),
                    node_ptr=self._node_ptr,
                    n_closest=h["n_closest"].copy(), coord_map=h["coord_map"].copy(), nedges=h["nedges"].copy(),
                    nsel=h["nsel"].copy())

    # ------------------------------------------------------------------
    def step(self, actions):
        """actions (B,) ints in [0, N]; returns (state, rewards (B,), dones (B,), infos)."""
        self.step_begin(actions)
        return self.step_end()

    def step_begin(self, actions):
        """First half of `step`: everything that is only ENQUEUED (vertex removal, smoothing, topology, interpolation,
        forces, the asynchronous read-back).  The caller may overlap other work (e.g. an optimiser step on another
        stream) with the GPU before `step_end` waits for the results."""
        B, N, h = self.B, self.N, self.h
        actions = np.asarray(actions).astype(np.int64)
        code = np.zeros(B, np.int32)  # 0 ok, 2 broken (Env2DAirfoil.py:342-364)
        shift = actions == N                                   # "do nothing": move the selection window
        pick = (actions >= 0) & (actions < h["nsel"]) & ~shift
        self.offset[shift] += 1
        rem = np.where(pick, h["coord_map"][np.arange(B), np.clip(actions, 0, N - 1)], -1).astype(np.int32)
        code[~shift & ~pick] = 2                               # KeyError in coord_map: "RAN OUT OF VERTICES"
        if self.gpu_remesh:
            dev, dt = self.device, self.dtopo
            rem_d = torch.from_numpy(rem).to(dev)
            dt.offset.copy_(torch.from_numpy(self.offset))
            remesh_batch_gpu(dt.coords, dt.cells, dt.nv, dt.nt, rem_d, self._rstat)
            its = torch.where((rem_d >= 0) & (self._rstat == 0), 50, 0).to(torch.int32)
            smooth_batch_gpu(dt.coords, dt.cells, dt.nv, dt.nt, its)
            # host mirrors of the meshes: device-to-host copies on a side stream, off the critical path of the step
            # (4 MB per 128 meshes; complete before the host logic below reads nv / status)
            pin = self.topo.pinned
            self._mirror_ev.record(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._mirror_stream):
                self._mirror_stream.wait_event(self._mirror_ev)
                pin["coords"].copy_(dt.coords, non_blocking=True)
                pin["cells"].copy_(dt.cells, non_blocking=True)
                pin["nv"].copy_(dt.nv, non_blocking=True)
                pin["nt"].copy_(dt.nt, non_blocking=True)
                self._rstat_host.copy_(self._rstat, non_blocking=True)
                self._mirror_done.record(self._mirror_stream)
            status = None
        elif self.gpu_smoothing:
            # host: cavity re-triangulation + Delaunay restoration only; GPU: smooth(50) of the changed meshes
            status = remesh_batch(self.coords, self.cells, self.nv, self.nt, rem, 0, self.nthreads)
            dev = self.device
            its = torch.from_numpy(np.where((rem >= 0) & (status == 0), 50, 0).astype(np.int32)).to(dev)
            if self.gpu_topology:
                self._upload_mesh()
                dt = self.dtopo
                smooth_batch_gpu(dt.coords, dt.cells, dt.nv, dt.nt, its)
                # the host engine needs the smoothed coordinates for its next cavity: asynchronous D2H into the
                # page-locked array, complete before this step's results are read back
                self.topo.pinned["coords"].copy_(dt.coords, non_blocking=True)
            else:
                up = self.topo.upload
                tc = up("coords", dev)
                smooth_batch_gpu(tc, up("cells", dev), up("nv", dev), up("nt", dev), its)
                self.topo.pinned["coords"].copy_(tc)    # D2H into the page-locked array (synchronises this stream)
        else:
            status = remesh_batch(self.coords, self.cells, self.nv, self.nt, rem, 50, self.nthreads)
        self._refresh_launch()
        self._step_pending = (code, status)

    def step_end(self):
        """Second half of `step`: wait for the results, rewards / terminal flags / in-place resets, next state."""
        B, N, h = self.B, self.N, self.h
        code, status = self._step_pending
        self._step_pending = None
        prev_flow = None
        if self.flow_overlap:     # results of the IPCS step launched in the PREVIOUS env step (this step's is still running)
            prev, self._flow_prev2 = getattr(self, "_flow_prev2", None), self._flow_prev
            if prev is not None:
                res = self._flow_res[prev]
                res["done"].synchronize()
                prev_flow = (res["host"][0].numpy().copy(), res["host"][1].numpy().copy())
        self._refresh_collect()
        if status is None:
            self._mirror_done.synchronize()
            status = self._rstat_host.numpy()
        code[status != 0] = 2
        code[h["nsel"] < N] = 2  # out of vertices
        rewards = np.zeros(B)
        dones = np.zeros(B, bool)
        drag_factor = -2 * np.log(0.5) / self.threshold
        err = np.abs(self.gt_drag[None] - self.new_drags) / np.abs(self.gt_drag[None])
        drag_reward = 2 * np.exp(-drag_factor * np.linalg.norm(err, axis=1)) - 1
        time_reward = (self.initial_num_node - self.nv) * self.TIME_REWARD
        acc = (np.abs(np.abs(self.gt_drag[None] - self.new_drags) / self.gt_drag[None]) > self.threshold).any(axis=1)
        vert = self.nv < self.goal_vertices * self.initial_num_node
        ok = code == 0
        rewards[ok] = (drag_reward + time_reward)[ok]
        dones[ok] = (acc | vert)[ok]
        rewards[~ok] = self.NEGATIVE_REWARD
        dones[~ok] = code[~ok] != 1        # (code 1 = "already removed": -1, not terminal, Env2DAirfoil.py:359-360; never produced)
        self.steps += 1
        dones |= self.steps >= self.timesteps
        infos = dict(code=code, nv=self.nv.copy(), new_drags=self.new_drags.copy(), new_lifts=self.new_lifts.copy())
        if self.flow_overlap:    # drag / lift of the re-solved flow of the PREVIOUS step's meshes (None at the first step)
            infos.update(flow_lag=1, flow_drag=None if prev_flow is None else prev_flow[0],
                         flow_lift=None if prev_flow is None else prev_flow[1])
        elif self.flow_steps > 0:  # drag / lift of the re-solved flow on the coarsened meshes (before any auto-reset)
            infos.update(flow_lag=0, flow_drag=self.flow_drag.copy(), flow_lift=self.flow_lift.copy())
        if self.auto_reset and dones.any():
            self._restore_initial(np.flatnonzero(dones))
        state = self.get_state()
        if self._deferred_mirror is not None:   # (off the critical path: the GPU is already working on the next state)
            self.coords[self._deferred_mirror] = self.x0
            self.cells[self._deferred_mirror] = self.cells0
            self._deferred_mirror = None
        return state, rewards, dones, infos


    # ------------------------------------------------------------------ device-resident rollout
    def _state_device(self):
        """`get_state` without the host: node features from device data only; the edge lists stay in the padded (B, EMAX)
        layout the topology engine writes (`mdq_gcn_forward_padded` and `mdq_replay_step` read them as they are: no edge
        offsets, no compaction launches)."""
        dev, B, N, S, dt = self.device, self.B, self.N, self.S, self.dtopo
        x = torch.empty((B, N, 2 + 3 * S), dtype=torch.float32, device=dev)
        _lib.check(self.lib.mdq_state_features(B, N, S, self.NV, self.NP, self._coords_dev.data_ptr(), self.u.data_ptr(),
                                               self.p.data_ptr(), dt.t["n_closest"].data_ptr(), dt.t["nsel"].data_ptr(),
                                               x.data_ptr(), _lib.stream_ptr()), "mdq_state_features")
        return dict(x=x, node_ptr=self._node_ptr, edge_src_pad=dt.t["edge_src"], edge_dst_pad=dt.t["edge_dst"],
                    nedges_dev=dt.t["nedges"])

    def rollout_device(self, fused, steps: int, explore=None, rand_actions=None, actions=None):
        """`steps` batched env steps WITHOUT a host round trip inside a step: the Q-network forward (`fused`: a
        `FusedGcn`), the epsilon-greedy choice (`explore` (steps,B) bool + `rand_actions` (steps,B) ints, drawn by the
        caller from its own random streams; greedy = first maximum of the Q-row) or given `actions` (steps,B), the
        action decoding, vertex removal, smoothing, topology, interpolation, forces, (S3: the IPCS step), reward /
        terminal logic and the in-place resets are all kernels on the current stream (`mdq_env_act`, `mdq_remesh`,
        `mdq_smooth_fast_env`, `mdq_env_topology`, ..., `mdq_env_result`, `mdq_restore_rows_masked`).
        Same semantics as `steps` calls of `step()` (tested against it).  Returns dict(rewards (steps,B), dones,
        actions, codes, nv) - read back ONCE at the end, when the host mirrors of the environments are refreshed too."""
        cur = torch.cuda.current_stream(self.device)
        if cur == torch.cuda.default_stream(self.device):
            # not on the legacy default stream: with the main chain there, the factorisation kernel of the flow stream
            # was measured to serialise with it (2.5 instead of 1.8 ms per batched step); a stream of the pool is fine
            if getattr(self, "_main_stream", None) is None:
                from .streams import role_streams
                self._main_stream = role_streams(self.device)["main"]
            self._main_stream.wait_stream(cur)
            with torch.cuda.stream(self._main_stream):
                out = self.rollout_device(fused, steps, explore, rand_actions, actions)
            cur.wait_stream(self._main_stream)
            return out
        ro = self.rollout_begin(steps, explore, rand_actions, actions)
        for k in range(int(steps)):
            self.rollout_step(ro, fused)
        return self.rollout_end(ro)

    def calibrate_streams(self, fused, tries: int = 6, steps: int = 8):
        """Pick a flow stream that REALLY runs beside the current (main) stream, by measurement.  HIP maps streams
        round-robin onto hardware queues; besides the pairs that land on one queue (the flow leg then runs behind the
        smoothing kernel: 3.5 ms per batched step instead of 1.85) there are pairs that overlap only partly (2.6 ms;
        two of eight candidates in `tools/time_stream_matrix.py`), and no synthetic probe tried separates those from the
        good ones.  So: a few real env steps with the current flow stream and with up to `tries - 1` fresh ones, the
        fastest stays.  The environments are reset afterwards (`reset_all`); call it on the stream the rollouts will run
        on, once, before they start.  Returns the measured ms per batched step of every candidate."""
        if not (self.flow_overlap and self.gpu_remesh):
            return []
        dev, B = self.device, self.B
        rng = np.random.default_rng(20251)
        cur = torch.cuda.current_stream(dev)

        def timed(k):
            ro = self.rollout_begin(k, rng.random((k, B)) < 0.5, rng.integers(0, self.N + 1, (k, B)))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for _ in range(k):
                self.rollout_step(ro, fused)
            e1.record(cur)
            self.rollout_end(ro)
            e1.synchronize()
            return e0.elapsed_time(e1) / k
        from . import streams as _st
        known = _st.calibrated_flow_stream(dev, cur)
        if known is not None:                       # this process has already chosen a flow stream for this main stream
            self.flow_wait()
            self._flow_stream = known
            self._calibrated_for = cur
            self.calibration_ms = []
            return []
        if (_st.roles_own_queues(dev) and self._flow_stream is _st.role_streams(dev)["flow"] and
                (cur is _st.role_streams(dev)["main"] or cur == _st.role_streams(dev)["main"] or _st._overlaps(self._flow_stream, cur, dev))):
            # CU-mask role streams with every probe passed: a hardware queue each, nothing to choose between.  (The probes of
            # `role_streams` compare the roles with the MAIN role: a caller on another stream - the bench's env groups - is
            # probed here, once, before the shortcut is taken)
            self._calibrated_for = cur
            self.calibration_ms = []
            _st.remember_flow_stream(dev, cur, self._flow_stream, [], "role streams own their hardware queues: no calibration")
            return []
        results = []
        how = "first two candidates agree"
        for t in range(max(1, int(tries))):
            if t > 0:
                self.flow_wait()
                self._flow_stream = _st.new_flow_candidate(dev)
            timed(3)
            results.append((timed(int(steps)), self._flow_stream))
            ms = [r[0] for r in results]
            if len(ms) == 2 and abs(ms[0] - ms[1]) <= 0.03 * min(ms):
                break                               # the role stream and ONE fresh stream agree: both overlap
            if len(ms) >= 2 and min(ms) < 0.85 * max(ms) and ms[-1] <= 1.03 * min(ms):
                how = "both behaviours seen"
                break                               # both behaviours seen and the current candidate is a good one
            if len(ms) > 2:
                how = "full calibration"
        best = min(results, key=lambda r: r[0])
        self.flow_wait()
        self._flow_stream = best[1]
        self._calibrated_for = cur
        self.calibration_ms = [r[0] for r in results]
        _st.remember_flow_stream(dev, cur, best[1], self.calibration_ms, how)
        self.reset_all()
        return self.calibration_ms

    def rollout_begin(self, steps: int, explore=None, rand_actions=None, actions=None):
        """First third of `rollout_device` (the learning loop interleaves its own launches with the steps): uploads the
        per-step action inputs, allocates the per-step outputs; `ro["state"]` is the current batched state on the device
        (x, packed and padded edge lists, `nedges`)."""
        if not self.gpu_remesh:
            raise _lib.MeshDQNHipError("rollout_device needs the device mesh engine (gpu_remesh=True)")
        dev, dt, B, K = self.device, self.dtopo, self.B, int(steps)
        i32 = torch.int32
        if self._pending is not None:
            self._refresh_collect()
        if actions is not None:
            act_all = torch.from_numpy(np.ascontiguousarray(actions, dtype=np.int32).reshape(K, B)).to(dev)
            expl_all = rand_all = None
        else:
            act_all = torch.empty((K, B), dtype=i32, device=dev)
            expl_all = torch.from_numpy(np.ascontiguousarray(explore, dtype=np.uint8).reshape(K, B)).to(dev)
            rand_all = torch.from_numpy(np.ascontiguousarray(rand_actions, dtype=np.int32).reshape(K, B)).to(dev)
        ro = dict(K=K, k=0, given=actions is not None, act=act_all, explore=expl_all, rand=rand_all,
                  rew=torch.empty((K, B), dtype=torch.float64, device=dev), done=torch.empty((K, B), dtype=torch.uint8, device=dev),
                  code=torch.empty((K, B), dtype=i32, device=dev), nv=torch.empty((K, B), dtype=i32, device=dev),
                  rem=torch.empty(B, dtype=i32, device=dev), code_act=torch.empty(B, dtype=i32, device=dev),
                  # step counters: read by every workgroup of mdq_env_finish, written to the other array (si: the current one)
                  d_steps=[torch.from_numpy(self.steps.astype(np.int32)).to(dev), torch.empty(B, dtype=i32, device=dev)], si=0,
                  err=torch.zeros(1, dtype=i32, device=dev))
        if getattr(self, "_gt_drag_dev", None) is None:
            self._gt_drag_dev = torch.from_numpy(np.ascontiguousarray(self.gt_drag, dtype=np.float64)).to(dev)
        dt.offset.copy_(torch.from_numpy(self.offset))
        ro["state"] = self._state_device()
        return ro

    def rollout_step(self, ro, fused, pack: bool = True):
        """One batched env step of a `rollout_begin` context, enqueued on the current stream; afterwards `ro["state"]`
        is the new batched state and `ro["act"][k] / ro["rew"][k] / ro["done"][k]` (device) describe the step (k = ro["k"] - 1).
        `pack=False`: the Q-forward uses the packed parameter copy as it is (a learning loop whose optimiser runs on another
        stream brings it up to date itself, at a point that is ordered against the parameter writes).
        Launches of a step (round 4: 8 + the hand-back launch of the smoothing, 14 in round 3): Q-forward (embedding, MLP head),
        `mdq_remesh_act` (action decoding + vertex removal), `mdq_smooth_fast_env`, `mdq_env_topology`,
        `mdq_interpolate_snapshots`, `mdq_probe_forces`, `mdq_env_finish` (reward / terminal logic, hand-over of the meshes
        to the flow stream, in-place resets, node features of the next state)."""
        dt, lib, B, N, S, k = self.dtopo, self.lib, self.B, self.N, self.S, ro["k"]
        if k >= ro["K"]:
            raise IndexError("rollout_step beyond the steps of rollout_begin")
        sp = _lib.stream_ptr
        st, rem = ro["state"], ro["rem"]
        q = None
        if not ro["given"]:
            q = fused.forward_arrays(st["x"], st["node_ptr"], st["edge_src_pad"], st["edge_dst_pad"], None, N, self.EMAX,
                                     pack=pack, edge_cnt=st["nedges_dev"])
        NVc, NTc = dt.coords.shape[1], dt.cells.shape[1]
        _lib.check(lib.mdq_remesh_act(B, NVc, NTc, dt.coords.data_ptr(), dt.cells.data_ptr(), dt.nv.data_ptr(), dt.nt.data_ptr(),
                                      N, None if q is None else q.data_ptr(),
                                      None if ro["explore"] is None else ro["explore"][k].data_ptr(),
                                      None if ro["rand"] is None else ro["rand"][k].data_ptr(), dt.t["nsel"].data_ptr(),
                                      dt.t["coord_map"].data_ptr(), dt.offset.data_ptr(), ro["act"][k].data_ptr(),
                                      rem.data_ptr(), ro["code_act"].data_ptr(), self._rstat.data_ptr(),
                                      *remesh_workspace(self.device, sp(), B, NVc, NTc), sp()), "mdq_remesh_act")
        tm = getattr(self, "smooth_events", None)     # (bench: HIP events around the launch, on this stream)
        if tm is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        # smooth(50) where a vertex was removed (mdq_env_smooth_iters folded into the smoothing launch)
        smooth_env_gpu(dt.coords, dt.cells, dt.nv, dt.nt, rem, self._rstat, 50)
        if tm is not None:
            e1.record()
            tm.append((e0, e1))
        flow = self.flow_steps > 0 and self.flow_overlap
        early = flow and self.gpu_topology and not self._late_handover
        sparse = 0 if os.environ.get("MDQ_FULL_INTERP", "") == "1" else (1 if self.flow_steps > 0 else 2)
        self._refresh_launch(readback=False, defer_flow=True, after_topology=self._flow_handover_mesh if early else None, sparse=sparse,
                             before_topology=self._flow_handover_target if early and self._handover_in_kernel else None)
        # ---- the end of the step in one launch
        x = torch.empty((B, N, 2 + 3 * S), dtype=torch.float32, device=self.device)
        d = self._finish_desc(ro, k, x)
        fin = None
        if flow and early:
            fin = self._flow_fin_early
            self._finish_handover(d, [(fin["u_n"], self.u[:, self.S - 1]), (fin["p_n"], self.p[:, self.S - 1])])
        elif flow:
            fin, pairs = self._flow_handover(self.u, self.p)
            self._finish_handover(d, pairs)
        _lib.check(lib.mdq_env_finish(C.byref(d), sp()), "mdq_env_finish")
        if fin is not None:
            self._flow_start(fin, mesh_early=early)
        ro["si"] ^= 1
        ro["state"] = dict(x=x, node_ptr=self._node_ptr, edge_src_pad=dt.t["edge_src"], edge_dst_pad=dt.t["edge_dst"],
                           nedges_dev=dt.t["nedges"])
        ro["k"] = k + 1

    def _finish_desc(self, ro, k, x):
        """Descriptor of `mdq_env_finish` for step k of a rollout (built once per environment object; the per-step
        pointers are patched in)."""
        dt, B, N, S = self.dtopo, self.B, self.N, self.S
        d = getattr(self, "_fin_desc", None)
        if d is None:
            d = self._fin_desc = _lib.EnvFinishDesc()
            d.B, d.N, d.S, d.NV, d.NP = B, N, S, self.NV, self.NP
            d.nv0, d.timesteps, d.auto_reset = int(self.initial_num_node), int(self.timesteps), 1 if self.auto_reset else 0
            d.threshold, d.time_reward, d.goal_vertices, d.negative_reward = (float(self.threshold), float(self.TIME_REWARD),
                                                                                float(self.goal_vertices), float(self.NEGATIVE_REWARD))
            d.gt_drag, d.nv, d.rstat = self._gt_drag_dev.data_ptr(), dt.nv.data_ptr(), self._rstat.data_ptr()
            d.topo_status, d.nsel, d.n_closest = dt.status.data_ptr(), dt.t["nsel"].data_ptr(), dt.t["n_closest"].data_ptr()
            self._fin_arrive = torch.zeros(B, dtype=torch.int32, device=self.device)   # (every launch leaves it at zero)
            d.arrive = self._fin_arrive.data_ptr()
        c = self._init_cache
        if c.get("x") is None:      # node features of the initial state (what an environment shows right after its reset)
            xi = torch.empty((1, N, 2 + 3 * S), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.mdq_state_features(1, N, S, self.NV, self.NP, self._x0_dev.data_ptr(), c["u"].data_ptr(),
                                                   c["p"].data_ptr(), c["dev"]["n_closest"].data_ptr(),
                                                   c["dev"]["nsel"].data_ptr(), xi.data_ptr(), _lib.stream_ptr()),
                       "mdq_state_features")
            c["x"] = xi
        d.x_init = c["x"].data_ptr()
        ra = self._restore_arg_arrays()                # (u / p / coords change their addresses from step to step)
        n = ra["n"]
        if n + 2 > _lib.FINISH_MAX_ROWS:
            raise _lib.MeshDQNHipError("mdq_env_finish: too many row arrays")
        d.n_rows = n
        for t in range(n):
            d.dst[t], d.src[t], d.row_bytes[t] = ra["dst"][t], ra["src"][t], ra["nbytes"][t]
            d.handover_dst[t], d.handover_off[t], d.handover_bytes[t] = None, 0, 0
        if not self.auto_reset:                        # nothing is restored: the rows stay as hand-over sources only
            for t in range(n):
                d.src[t] = None
        else:
            # the interpolated snapshots (rows 0 / 1: 0.3 MB per environment) are NOT restored here: every step of a rollout
            # interpolates them again on the current meshes before anything reads them, and the state of a reset environment
            # comes from the cached features (x_init); they stay hand-over sources (the warm start of the flow leg)
            d.src[0] = d.src[1] = None
        d.new_drags = self._dev_drag.data_ptr()
        d.code_in, d.code_out = ro["code_act"].data_ptr(), ro["code"][k].data_ptr()
        d.steps_in, d.steps_out = ro["d_steps"][ro["si"]].data_ptr(), ro["d_steps"][ro["si"] ^ 1].data_ptr()
        d.reward, d.done, d.err_flag, d.nv_out = (ro["rew"][k].data_ptr(), ro["done"][k].data_ptr(), ro["err"].data_ptr(),
                                                  ro["nv"][k].data_ptr())
        d.coords, d.u, d.p, d.x = self._coords_dev.data_ptr(), self.u.data_ptr(), self.p.data_ptr(), x.data_ptr()
        return d

    def _finish_handover(self, d, pairs):
        """The hand-over windows of `mdq_env_finish`: every (destination, source) pair of `_flow_handover` is a window of
        one of the row arrays (a whole row, or - the warm start - the last snapshot of the u / p rows) or, for arrays that
        are not reset in place (the edge numbering), a row array of its own without a source."""
        n = d.n_rows
        index = {int(d.dst[t]): t for t in range(n)}
        for dst, src in pairs:
            if dst.dtype != src.dtype or not dst.is_contiguous() or dst.shape != src.shape or not src[0].is_contiguous():
                raise ValueError("flow hand-over: buffers of different shapes")
            base = src._base if src._base is not None else src
            per = src[0].numel() * src.element_size()
            t = index.get(int(base.data_ptr()))
            if t is None:                              # not one of the restored arrays: hand-over only
                if not src.is_contiguous():
                    raise ValueError("flow hand-over: an array that is not reset must be contiguous")
                t = n
                n += 1
                d.dst[t], d.src[t], d.row_bytes[t] = src.data_ptr(), None, per
                off = 0
            else:
                off = src.data_ptr() - base.data_ptr()
                if off < 0 or off + per > d.row_bytes[t] or (not src.is_contiguous() and src.stride(0) * src.element_size() != d.row_bytes[t]):
                    raise ValueError("flow hand-over: the source is not a window of a row array")
            d.handover_dst[t], d.handover_off[t], d.handover_bytes[t] = dst.data_ptr(), off, per
        d.n_rows = n

    def rollout_end(self, ro):
        """The one read-back of a rollout; the host mirrors of the environments follow the device."""
        K = ro["k"]
        last = getattr(self, "_interp_last", None)
        if K and last is not None and last[1]:
            # the steps of a rollout interpolate only what they read (sparse: most edge-midpoint entries of `self.u` are skipped);
            # `self.u` / `self.p` are public - whoever reads them after the rollout (field dumps, deploy, tests, a host-driven
            # step()) finds the COMPLETE fields of the last step's meshes: one full pass here, once per rollout (~45 us)
            d = last[0]
            d.sparse = 0
            _lib.check(self.lib.mdq_interpolate_snapshots(C.byref(d), _lib.stream_ptr()), "mdq_interpolate_snapshots")
            self._interp_last = (d, 0, last[2])
        # ONE read-back: the per-step outputs + every small host mirror as one packed buffer (a dozen device-to-host copies, each
        # a synchronisation of its own, were 0.4-0.5 ms per rollout: round 4's `rollout_end` was 0.7-0.95 ms - 5 % of the
        # driver's 20-step rollouts), the meshes (4 MB per 128 environments) into their page-locked mirrors in front of it
        dev, dt, h = self.device, self.dtopo, self.h
        main = torch.cuda.current_stream(dev)
        pin = self.topo.pinned
        # (on the MAIN stream: the side stream of the host-driven step() - a plain pool stream - cost the S3 rollouts of a process
        #  40 % once it had been used here, 0.73 -> 1.04-1.10 ms per step in bench.py: the stream -> hardware-queue lottery of
        #  HISTORY 5 "Streams"; the copies run behind the last step's kernels either way)
        pin["coords"].copy_(dt.coords, non_blocking=True)
        pin["cells"].copy_(dt.cells, non_blocking=True)
        parts = [("rewards", ro["rew"][:K]), ("dones", ro["done"][:K]), ("actions", ro["act"][:K]), ("codes", ro["code"][:K]),
                 ("nv_steps", ro["nv"][:K]), ("err", ro["err"]), ("nv", dt.nv), ("nt", dt.nt), ("offset", dt.offset),
                 ("steps", ro["d_steps"][ro["si"]]), ("drag", self._dev_drag), ("lift", self._dev_lift)]
        parts += [(k, dt.t[k]) for k in ("nsel", "nedges", "ne", "coord_map", "n_closest")]
        if self.flow_steps > 0:
            parts.append(("flow_status", self.flow_status))        # (legs up to the previous step: the last one may still run)
        flat = [t.contiguous().view(torch.uint8).reshape(-1) for _, t in parts]
        pad = [(-f.numel()) % 8 for f in flat]                  # (every part starts 8-byte aligned in the packed buffer)
        packed = torch.cat([x for f, p_ in zip(flat, pad) for x in ((f, f.new_zeros(p_)) if p_ else (f,))])
        host = getattr(self, "_rollout_host", None)
        if host is None or host.numel() < packed.numel():
            host = self._rollout_host = torch.empty(max(packed.numel(), 1 << 16), dtype=torch.uint8, pin_memory=True)
        host[:packed.numel()].copy_(packed, non_blocking=True)
        self._packed_ev.record(main)
        self._packed_ev.synchronize()
        buf, off, got = host.numpy(), 0, {}
        for (name, t), f, p_ in zip(parts, flat, pad):
            np_dt = {torch.float64: np.float64, torch.int32: np.int32, torch.uint8: np.uint8, torch.int64: np.int64}[t.dtype]
            got[name] = buf[off:off + f.numel()].view(np_dt).reshape(tuple(t.shape)).copy()
            off += f.numel() + p_
        out = dict(rewards=got["rewards"], dones=got["dones"].astype(bool), actions=got["actions"], codes=got["codes"],
                   nv=got["nv_steps"])
        if int(got["err"][0]) != 0:
            raise _lib.MeshDQNHipError("topology kernel failed inside rollout_device")
        if K:               # a failed flow leg is an ERROR here, not a NaN in what the caller reads later
            _check_flow_forces(got["drag"], got["lift"], "rollout_device", got.get("flow_status"))
        self.nv[...], self.nt[...], self.offset[...], self.steps[...] = got["nv"], got["nt"], got["offset"], got["steps"]
        for k in ("nsel", "nedges", "ne", "coord_map", "n_closest"):
            h[k][...] = got[k]
        self.new_drags, self.new_lifts = got["drag"], got["lift"]
        last_done = out["dones"][-1] if K and self.auto_reset else None
        if last_done is not None and last_done.any():      # (restarted environments: the cached initial forces, like step())
            self.new_drags[last_done] = self._init_cache["drags"]
            self.new_lifts[last_done] = self._init_cache["lifts"]
            # ... and their interpolated snapshots: `mdq_env_finish` leaves them alone inside a rollout (every step
            # interpolates again before anything reads them); whoever reads the state after the LAST step - get_state(), a
            # host-driven step() - must find the initial fields in the rows of the environments that step reset
            c = self._init_cache
            ti = torch.from_numpy(np.flatnonzero(last_done).astype(np.int32)).to(self.device)
            dst = (C.c_void_p * 2)(self.u.data_ptr(), self.p.data_ptr())
            src = (C.c_void_p * 2)(c["u"].data_ptr(), c["p"].data_ptr())
            nb = (C.c_int64 * 2)(self.u[0].numel() * 8, self.p[0].numel() * 8)
            _lib.check(self.lib.mdq_restore_rows(2, dst, src, nb, int(ti.numel()), ti.data_ptr(), _lib.stream_ptr()),
                       "mdq_restore_rows")
        self._deferred_mirror = None
        return out


def _check_flow_forces(drag, lift, where, status=None):
    """Raises when a flow leg reported a step it could not take: a set status word (mdq_ipcs_desc.status: the two-workgroup
    operator modes 4 / 7 abandon a step when a team barrier times out - the partner workgroup was not resident because another
    process or stream held its CU; csrc/mdq_ipcs.hip `team_failed`; u_n / p_n of that environment were not advanced) or
    non-finite drag / lift (the same event seen through the forces, or a solve that diverged)."""
    bad = ~(np.isfinite(drag) & np.isfinite(lift))
    bad = bad.reshape(bad.shape[0], -1).any(1)
    timed_out = np.zeros_like(bad) if status is None else (np.asarray(status) != 0)
    if bad.any() or timed_out.any():
        which = np.flatnonzero(bad | timed_out)
        raise _lib.MeshDQNHipError(
            f"{where}: the flow leg failed in environment(s) {which[:8].tolist()}{' ...' if which.size > 8 else ''}: "
            + ("team barrier time-out of the two-workgroup IPCS modes (is another process or a CU-masked stream holding CUs? "
               "MDQ_NO_TEAM_TILES=1 keeps one workgroup per environment)" if timed_out.any() else "non-finite drag / lift"))


class VecEnvGroups:
    """`num_envs` environments as G independent `VecEnv2DAirfoil` groups, each driven by its own Python thread on
    its own HIP stream - the counterpart of the reference's asynchronous Ray workers (`num_parallel`,
    airfoil_dqn.py:428-503) inside one process: while one group waits for its GPU kernels (the smoothing kernel is
    latency-bound and occupies only as many CUs as the group has environments) another group runs its host mesh
    engine calls (ctypes and torch release the GIL).  All groups share one base environment (ground truth,
    snapshots, interpolation grid)."""

    def __init__(self, config, num_envs: int, groups: int = 2, compute_device="cuda", base_env: Env2DAirfoil | None = None, **kw):
        self.device = torch.device(compute_device)
        base = base_env or Env2DAirfoil(config, compute_device=compute_device)
        G = max(1, min(int(groups), int(num_envs)))
        sizes = [num_envs // G + (1 if g < num_envs % G else 0) for g in range(G)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(G)]
        self.envs = []
        for g in range(G):
            with torch.cuda.stream(self.streams[g]):
                self.envs.append(VecEnv2DAirfoil(config, sizes[g], compute_device=compute_device, base_env=base, **kw))
        torch.cuda.synchronize(self.device)
        self.B = int(num_envs)

    def rollout(self, act_fn, steps: int, barrier=None):
        """Every group runs `steps` batched steps concurrently: state -> act_fn(group, env, state) -> env.step.
        Returns the list of per-group (rewards, dones) of the last step.  Exceptions of the workers are re-raised."""
        import threading
        out, errs = [None] * len(self.envs), []

        def work(g):
            try:
                env = self.envs[g]
                with torch.cuda.stream(self.streams[g]):
                    st = env.get_state()
                    for _ in range(steps):
                        st, rew, done, info = env.step(act_fn(g, env, st))
                    self.streams[g].synchronize()
                    out[g] = (rew, done)
            except BaseException as exc:  # noqa: BLE001 - surfaced to the caller below
                errs.append(exc)

        threads = [threading.Thread(target=work, args=(g,)) for g in range(len(self.envs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise errs[0]
        return out

    def rollout_device(self, fused, steps: int, explore, rand_actions):
        """Every group runs `VecEnv2DAirfoil.rollout_device` (no host round trip inside a step) on its own stream:
        fused[g] / explore[g] / rand_actions[g] per group.  Returns the per-group result dicts."""
        import threading
        out, errs = [None] * len(self.envs), []

        def work(g):
            try:
                with torch.cuda.stream(self.streams[g]):
                    out[g] = self.envs[g].rollout_device(fused[g], steps, explore[g], rand_actions[g])
                    self.streams[g].synchronize()
            except BaseException as exc:  # noqa: BLE001 - surfaced to the caller below
                errs.append(exc)

        if len(self.envs) == 1:
            work(0)
        else:
            threads = [threading.Thread(target=work, args=(g,)) for g in range(len(self.envs))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if errs:
            raise errs[0]
        return out
