"""ctypes binding of `libmeshdqn_hip.so` (the C ABI declared in include/meshdqn_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call
fails, a `MeshDQNHipError` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (MDQ_LIB_PATH: a differently built copy of the library, e.g. an experiment of tools/: development knob)
LIB_PATH = os.environ.get("MDQ_LIB_PATH") or os.path.join(HERE, "libmeshdqn_hip.so")

ABI_VERSION = 7


class MeshDQNHipError(RuntimeError):
    pass


class IpcsDesc(C.Structure):
    """Mirror of `mdq_ipcs_desc` (include/meshdqn_hip.h) - keep field order in sync."""
    _fields_ = [
        ("B", C.c_int32), ("NV", C.c_int32), ("NT", C.c_int32), ("NE", C.c_int32),
        ("N2", C.c_int32), ("NNZ2", C.c_int32), ("NNZ1", C.c_int32), ("NAF", C.c_int32),
        ("NSE2", C.c_int32), ("NSE1", C.c_int32),
        ("mu", C.c_double), ("rho", C.c_double), ("dt", C.c_double),
        ("rtol", C.c_double),
        ("maxit_u", C.c_int32), ("maxit_p", C.c_int32), ("maxit_m", C.c_int32), ("mode", C.c_int32),
        ("nv", C.c_void_p), ("nt", C.c_void_p), ("ne", C.c_void_p), ("naf", C.c_void_p),
        ("coords", C.c_void_p), ("cell_dofs", C.c_void_p), ("cell_outflow", C.c_void_p),
        ("rowptr2", C.c_void_p), ("colidx2", C.c_void_p), ("asm2_ptr", C.c_void_p), ("asm2_src", C.c_void_p),
        ("rowptr1", C.c_void_p), ("colidx1", C.c_void_p), ("asm1_ptr", C.c_void_p), ("asm1_src", C.c_void_p),
        ("sl2_off", C.c_void_p), ("sl2_col", C.c_void_p), ("sl1_off", C.c_void_p), ("sl1_col", C.c_void_p),
        ("mf_scat", C.c_void_p), ("mf_tptr", C.c_void_p),
        ("g2_ptr", C.c_void_p), ("g2_src", C.c_void_p), ("g1_ptr", C.c_void_p), ("g1_src", C.c_void_p),
        ("bcu_flag", C.c_void_p), ("bcu_gx", C.c_void_p), ("bcp_flag", C.c_void_p),
        ("NBO", C.c_int32), ("NBE", C.c_int32),
        ("nbo", C.c_void_p), ("bo_rows", C.c_void_p), ("bo_ptr", C.c_void_p), ("bo_col", C.c_void_p),
        ("bo_src", C.c_void_p), ("bo_val", C.c_void_p),
        ("pd_enabled", C.c_int32), ("NPART", C.c_int32), ("NPW", C.c_int32), ("NPF", C.c_int32),
        ("NPGI", C.c_int32), ("NPS", C.c_int32), ("NPGK", C.c_int32), ("pcg_degree", C.c_int32),
        ("pd_hdr", C.c_void_p), ("pd_node", C.c_void_p), ("pd_meta", C.c_void_p), ("pd_rowblk", C.c_void_p),
        ("pd_W", C.c_void_p), ("pd_F", C.c_void_p), ("pd_gidx", C.c_void_p), ("pd_Sinv", C.c_void_p),
        ("pd_gk_ptr", C.c_void_p), ("pd_gk_col", C.c_void_p), ("pd_gk_val", C.c_void_p),
        ("af_facets", C.c_void_p),
        ("geom", C.c_void_p), ("A1", C.c_void_p), ("Ms", C.c_void_p), ("K1s", C.c_void_p),
        ("lift1", C.c_void_p), ("lift3", C.c_void_p), ("idiag1", C.c_void_p),
        ("sdiagM", C.c_void_p), ("sdiagK", C.c_void_p),
        ("u_n", C.c_void_p), ("p_n", C.c_void_p),
        ("work", C.c_void_p), ("work_doubles", C.c_int64),
        ("mf_rlist", C.c_void_p), ("mf_rcnt", C.c_void_p), ("mf_lpos", C.c_void_p), ("NRL", C.c_int32), ("rl_flags", C.c_int32),
        ("status", C.c_void_p),
    ]


class InterpDesc(C.Structure):
    """Mirror of `mdq_interp_desc`."""
    _fields_ = [
        ("B", C.c_int32), ("S", C.c_int32), ("NP", C.c_int32), ("NP1", C.c_int32),
        ("src_nv", C.c_int32), ("src_nt", C.c_int32), ("src_n2", C.c_int32),
        ("gnx", C.c_int32), ("gny", C.c_int32), ("_pad", C.c_int32),
        ("x0", C.c_double), ("y0", C.c_double), ("inv_hx", C.c_double), ("inv_hy", C.c_double),
        ("npts", C.c_void_p), ("np1", C.c_void_p), ("points", C.c_void_p),
        ("src_coords", C.c_void_p), ("src_cell_dofs", C.c_void_p), ("src_geom", C.c_void_p),
        ("bin_ptr", C.c_void_p), ("bin_cells", C.c_void_p), ("src_u", C.c_void_p), ("src_p", C.c_void_p),
        ("out_u", C.c_void_p), ("out_p", C.c_void_p), ("out_cell", C.c_void_p), ("src_cellrec", C.c_void_p),
        ("npts_extra", C.c_void_p),
        ("af_facets", C.c_void_p), ("naf", C.c_void_p), ("cell_dofs", C.c_void_p),
        ("NT", C.c_int32), ("NAF", C.c_int32), ("sparse", C.c_int32), ("_pad2", C.c_int32),
    ]


class EnvTopoDesc(C.Structure):
    """Mirror of `mdq_env_topo_desc`."""
    _fields_ = [(n, C.c_int32) for n in ("B", "NV", "NT", "NP", "NAF", "N", "EMAX", "npoly")] + [
        (n, C.c_void_p) for n in ("coords", "cells", "nv", "nt", "offset", "polygon", "ne", "cell_dofs", "points", "naf",
                                  "af_facets", "nremovable", "nsel", "n_closest", "coord_map", "nedges", "edge_src",
                                  "edge_dst", "edge_len", "ipcs", "handover", "workspace")] + [("workspace_bytes", C.c_int64)]


class TopoHandover(C.Structure):
    """Mirror of `mdq_topo_handover`."""
    _fields_ = [(n, C.c_void_p) for n in ("coords", "cells", "nv", "nt", "cell_dofs", "ne")]


class IpcsTopoOut(C.Structure):
    """Mirror of `mdq_ipcs_topo_out`."""
    _fields_ = [(n, C.c_int32) for n in ("NBO", "NBE", "NSE1", "flow_only")] + [
        (n, C.c_void_p) for n in ("mf_scat", "cell_outflow", "bcu_flag", "bcu_gx", "bcp_flag", "nbo", "bo_rows", "bo_ptr",
                                  "bo_col", "bo_src", "g1_ptr", "g1_src", "g2_ptr", "g2_src", "sl1_off", "sl1_col",
                                  "cell_dofs_in", "ne_in")]


FINISH_MAX_ROWS = 16


class EnvFinishDesc(C.Structure):
    """Mirror of `mdq_env_finish_desc`."""
    _fields_ = [(n, C.c_int32) for n in ("B", "N", "S", "NV", "NP", "n_rows", "nv0", "timesteps", "auto_reset", "_pad")] + [
        (n, C.c_double) for n in ("threshold", "time_reward", "goal_vertices", "negative_reward")] + [
        (n, C.c_void_p) for n in ("new_drags", "gt_drag", "nv", "rstat", "topo_status", "nsel", "code_in", "code_out",
                                  "steps_in", "steps_out", "reward", "done", "err_flag", "nv_out")] + [
        ("dst", C.c_void_p * FINISH_MAX_ROWS), ("src", C.c_void_p * FINISH_MAX_ROWS),
        ("row_bytes", C.c_int64 * FINISH_MAX_ROWS), ("handover_dst", C.c_void_p * FINISH_MAX_ROWS),
        ("handover_off", C.c_int64 * FINISH_MAX_ROWS), ("handover_bytes", C.c_int64 * FINISH_MAX_ROWS)] + [
        (n, C.c_void_p) for n in ("coords", "u", "p", "n_closest", "x_init", "x", "arrive")]


# every symbol include/meshdqn_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mdq_abi_version": (C.c_int, []),
    "mdq_last_error": (C.c_char_p, []),
    "mdq_ipcs_workspace_doubles": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mdq_ipcs_assemble": (C.c_int, [C.POINTER(IpcsDesc), C.c_void_p]),
    "mdq_ipcs_setup_matfree": (C.c_int, [C.POINTER(IpcsDesc), C.c_void_p]),
    "mdq_ipcs_build_tile_maps": (C.c_int, [C.POINTER(IpcsDesc), C.c_void_p, C.c_void_p]),
    "mdq_flow_sort_cells": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "mdq_ipcs_evolve": (C.c_int, [C.POINTER(IpcsDesc), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_ipcs_evolve_timed": (C.c_int, [C.POINTER(IpcsDesc), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_double)]),
    "mdq_probe_forces": (C.c_int, [C.POINTER(IpcsDesc), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "mdq_gcn_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_gcn_forward_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_gcn_forward_padded": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_interpolate_snapshots": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mdq_remesh_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "mdq_remesh": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mdq_remesh_host": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "mdq_env_topology_host": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mdq_env_topology_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "mdq_env_topology": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_state_features": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_compact_edges": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "mdq_restore_rows": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mdq_env_act": (C.c_int, [C.c_int32, C.c_int32] + [C.c_void_p] * 10),
    "mdq_env_smooth_iters": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mdq_env_result": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int32,
                                 C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_restore_rows_masked": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mdq_env_finish": (C.c_int, [C.POINTER(EnvFinishDesc), C.c_void_p]),
    "mdq_remesh_act": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
                       + [C.c_void_p] * 11 + [C.c_int64, C.c_void_p]),
    "mdq_edge_ptr": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_smooth_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "mdq_smooth": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int64, C.c_void_p]),
    "mdq_smooth_stats": (C.c_int, [C.c_void_p, C.c_int32]),
    "mdq_smooth_fast_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "mdq_smooth_fast": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int64, C.c_void_p]),
    "mdq_smooth_fast_env": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "mdq_ipcs_factorize_pressure": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_ipcs_reset_history": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_gcn_train_workspace": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "mdq_gcn_train_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_gcn_pack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mdq_replay_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_replay_sample": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mdq_adam_step": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mdq_spin": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "mdq_stream_create_cu_mask": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mdq_stream_destroy": (C.c_int, [C.c_void_p]),
    "mdq_copy_strided": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mdq_smooth_host": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
}

_lib = None


def load():
    """Load the HIP library (building nothing: see meshdqn_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64; import it FIRST so that our library binds to the
    # same HIP runtime instance (two runtimes in one process do not share devices/streams)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MeshDQNHipError(
            f"{LIB_PATH} is missing - build it with `python -m meshdqn_amd.build` "
            "(the MI355X path has no CPU fallback)")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # pragma: no cover
        raise MeshDQNHipError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise MeshDQNHipError(f"{LIB_PATH} does not export {name}") from exc
        fn.restype = res
        fn.argtypes = args
    v = lib.mdq_abi_version()
    if v != ABI_VERSION:
        raise MeshDQNHipError(f"ABI version mismatch: library {v}, python {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mdq_last_error()
        raise MeshDQNHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def stream_ptr(stream=None):
    """hipStream_t of a torch stream (or of torch's current stream)."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)
