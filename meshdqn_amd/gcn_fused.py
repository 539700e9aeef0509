"""Fused HIP inference of `NodeRemovalNet` / `AirfoilGCNN` (host side of `mdq_gcn_forward`).

Packs the module's parameters (transposed to [in][out] so that consecutive lanes read consecutive
output channels), builds the per-graph node / edge ranges of a `Batch` and launches the two
kernels of meshdqn_amd/csrc/mdq_gcn.hip (inference) or the learning step of mdq_gcn_train.hip
(`FusedGcn.train_step`: forward + loss + backward without autograd).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib


class GcnLevel(C.Structure):
    _fields_ = [("type", C.c_int32), ("fin", C.c_int32), ("w_l", C.c_void_p), ("b", C.c_void_p),
                ("w_r", C.c_void_p), ("pool_w", C.c_void_p)]


class GcnNet(C.Structure):
    _fields_ = [("nlevels", C.c_int32), ("C", C.c_int32), ("fin0", C.c_int32), ("out_dim", C.c_int32),
                ("ratio", C.c_double), ("softmax", C.c_int32), ("_pad", C.c_int32),
                ("levels", GcnLevel * 6),
                ("lin1_w", C.c_void_p), ("lin1_b", C.c_void_p), ("lin2_w", C.c_void_p), ("lin2_b", C.c_void_p),
                ("lin3_w", C.c_void_p), ("lin3_b", C.c_void_p)]


def _levels_of(net):
    from .airfoilgcnn import AirfoilGCNN, NodeRemovalNet
    if isinstance(net, NodeRemovalNet):
        # conv3/pool3 and conv6/pool6 are skipped by the reference's forward (airfoilgcnn.py:106-110,124-128)
        return [(net.conv1, net.pool1), (net.conv2, net.pool2), (net.conv4, net.pool4), (net.conv5, net.pool5)], True
    if isinstance(net, AirfoilGCNN):
        return [(net.conv1, net.pool1), (net.conv2, net.pool2), (net.conv3, net.pool3),
                (net.conv4, net.pool4), (net.conv5, net.pool5), (net.conv6, net.pool6)], False
    raise TypeError(type(net))


class GcnGradLayout(C.Structure):
    _fields_ = [("w_l", C.c_int32 * 6), ("b", C.c_int32 * 6), ("w_r", C.c_int32 * 6), ("pool_w", C.c_int32 * 6),
                ("lin1_w", C.c_int32), ("lin1_b", C.c_int32), ("lin2_w", C.c_int32), ("lin2_b", C.c_int32),
                ("lin3_w", C.c_int32), ("lin3_b", C.c_int32), ("total", C.c_int32), ("_pad", C.c_int32)]


class GcnTrainDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("NMAX", C.c_int32), ("EMAX", C.c_int32), ("mode", C.c_int32), ("gamma", C.c_double),
                ("x", C.c_void_p), ("node_ptr", C.c_void_p), ("esrc", C.c_void_p), ("edst", C.c_void_p),
                ("edge_ptr", C.c_void_p), ("q_other", C.c_void_p), ("action", C.c_void_p), ("reward", C.c_void_p),
                ("nonfinal", C.c_void_p), ("workspace", C.c_void_p), ("partial", C.c_void_p), ("grad", C.c_void_p),
                ("loss", C.c_void_p), ("out", C.c_void_p), ("layout", GcnGradLayout)]


PACK_MAX = 32


class GcnPackTable(C.Structure):
    _fields_ = [("n", C.c_int32), ("_pad", C.c_int32), ("src", C.c_void_p * PACK_MAX), ("dst", C.c_void_p * PACK_MAX),
                ("rows", C.c_int32 * PACK_MAX), ("cols", C.c_int32 * PACK_MAX)]


class FusedGcn:
    """Device-side packed copy of a network's parameters + launchers (inference forward, learning step)."""

    def __init__(self, net):
        self.net = net
        self.lib = _lib.load()
        self._version = None
        self._table_key = None
        self._keep = []
        self.desc = None
        self._train_bufs = {}

    def _build_table(self):
        """Packed parameter buffers ([in][out] layout), the kernel descriptor pointing at them, the table of the
        one-launch repack (mdq_gcn_pack) and the offsets of the parameters inside the flat gradient."""
        net = self.net
        levels, softmax = _levels_of(net)
        dev = net.lin1.weight.device
        if dev.type != "cuda":
            raise _lib.MeshDQNHipError("fused GCN kernels need the module on a GPU (no CPU fallback)")
        keep, table = [], GcnPackTable()
        offs, pos = {}, 0
        for prm in net.parameters():
            offs[id(prm)] = pos
            pos += prm.numel()
        lay = GcnGradLayout()
        lay.total = pos

        def seg(prm, transpose=False):
            if prm.dtype != torch.float32 or not prm.is_contiguous():
                raise ValueError("fused GCN kernels need contiguous float32 parameters")
            buf = torch.empty(prm.numel(), dtype=torch.float32, device=dev)
            keep.append(buf)
            i = table.n
            if i >= PACK_MAX:
                raise ValueError("too many parameter segments")
            table.src[i], table.dst[i] = prm.data_ptr(), buf.data_ptr()
            if transpose:
                table.rows[i], table.cols[i] = prm.shape[0], prm.shape[1]
            else:
                table.rows[i], table.cols[i] = prm.numel(), 1
            table.n = i + 1
            return buf.data_ptr()

        d = GcnNet()
        d.nlevels = len(levels)
        d.C = net.lin1.weight.shape[1] // 2
        d.out_dim = net.lin3.weight.shape[0]
        d.ratio = float(levels[0][1].ratio)
        d.softmax = 1 if softmax else 0
        for l in range(6):
            lay.w_l[l] = lay.b[l] = lay.w_r[l] = lay.pool_w[l] = -1
        for l, (conv, pool) in enumerate(levels):
            lv = d.levels[l]
            if hasattr(conv, "lin_l"):
                lv.type = 0
                lv.fin = conv.lin_l.weight.shape[1]
                lv.w_l = seg(conv.lin_l.weight, True)
                lv.b = seg(conv.lin_l.bias)
                lv.w_r = seg(conv.lin_r.weight, True)
                lay.w_l[l], lay.b[l], lay.w_r[l] = offs[id(conv.lin_l.weight)], offs[id(conv.lin_l.bias)], offs[id(conv.lin_r.weight)]
            else:
                lv.type = 1
                lv.fin = conv.lin.weight.shape[1]
                lv.w_l = seg(conv.lin.weight, True)
                lv.b = seg(conv.bias)
                lv.w_r = None
                lay.w_l[l], lay.b[l] = offs[id(conv.lin.weight)], offs[id(conv.bias)]
            lv.pool_w = seg(pool.weight)
            lay.pool_w[l] = offs[id(pool.weight)]
            if l > 0 and lv.fin != d.C:
                raise ValueError("inner levels must have conv_width inputs")
        d.fin0 = d.levels[0].fin
        d.lin1_w, d.lin1_b = seg(net.lin1.weight, True), seg(net.lin1.bias)
        d.lin2_w, d.lin2_b = seg(net.lin2.weight, True), seg(net.lin2.bias)
        d.lin3_w, d.lin3_b = seg(net.lin3.weight, True), seg(net.lin3.bias)
        lay.lin1_w, lay.lin1_b = offs[id(net.lin1.weight)], offs[id(net.lin1.bias)]
        lay.lin2_w, lay.lin2_b = offs[id(net.lin2.weight)], offs[id(net.lin2.bias)]
        lay.lin3_w, lay.lin3_b = offs[id(net.lin3.weight)], offs[id(net.lin3.bias)]
        if net.lin1.weight.shape[0] != 128 or net.lin2.weight.shape[0] != 64:
            raise ValueError("head must be 2C -> 128 -> 64 -> out (as in the reference)")
        self.desc, self._keep, self._table, self.layout = d, keep, table, lay
        self._train_bufs = {}

    def _pack(self, stream=None):
        """Bring the packed copy up to date: ONE launch (mdq_gcn_pack), and only after the parameters changed."""
        net = self.net
        # (`_mdq_version`: bumped by whoever writes the parameters with a kernel of its own, e.g. mdq_adam_step)
        version = tuple(p._version for p in net.parameters()) + (id(net.conv1), getattr(net, "_mdq_version", 0))
        if version == self._version:
            return
        key = tuple(p.data_ptr() for p in net.parameters())
        if key != self._table_key:
            self._build_table()
            self._table_key = key
        _lib.check(self.lib.mdq_gcn_pack(C.byref(self._table), _lib.stream_ptr(stream)), "mdq_gcn_pack")
        self._version = version

    def train_step(self, x, node_ptr, esrc, edst, edge_ptr, nmax, emax, mode, q_other, action, reward, nonfinal, gamma,
                   loss_out=None, want_out=False, stream=None):
        """Forward + double-DQN Huber loss + backward of the module over a minibatch on the hand-written kernels
        (`mdq_gcn_train_step`, no autograd): returns (loss (1,) tensor, flat gradient over ALL parameters in
        `parameters()` order [, head outputs (B, out)]).  The returned tensors are persistent buffers that the next call
        overwrites.  mode 0: this network evaluates the states, `q_other` the next states; mode 1 the other way
        round (the reference's `select` toggle, airfoil_dqn.py:240-310).  `loss_out`: a (1,) float32 device tensor to
        receive the loss instead of the internal one (e.g. a slot of a log ring: no copy, no synchronisation)."""
        self._pack(stream)
        d = self.desc
        dev = x.device
        B = node_ptr.numel() - 1
        key = (B, int(nmax), int(emax))
        bufs = self._train_bufs.get(key)
        if bufs is None:
            ws = int(self.lib.mdq_gcn_train_workspace(C.byref(d), int(nmax), max(int(emax), 1)))
            if ws <= 0:
                raise _lib.MeshDQNHipError("mdq_gcn_train_workspace failed")
            bufs = dict(ws=torch.zeros((B, ws), dtype=torch.float32, device=dev),
                        partial=torch.zeros((B, self.layout.total), dtype=torch.float32, device=dev),
                        grad=torch.zeros(self.layout.total, dtype=torch.float32, device=dev),
                        loss=torch.zeros(1, dtype=torch.float32, device=dev),
                        out=torch.zeros((B, d.out_dim), dtype=torch.float32, device=dev))
            self._train_bufs = {key: bufs}
        x = x.reshape(-1, x.shape[-1])
        for t, dt in ((x, torch.float32), (q_other, torch.float32), (reward, torch.float32), (nonfinal, torch.float32),
                      (action, torch.int64), (node_ptr, torch.int32), (edge_ptr, torch.int32), (esrc, torch.int32),
                      (edst, torch.int32)):
            if t.dtype != dt or not t.is_contiguous() or t.device != dev:
                raise ValueError(f"train_step wants contiguous {dt} tensors on {dev}")
        if q_other.shape != (B, d.out_dim) or action.numel() != B or reward.numel() != B or nonfinal.numel() != B:
            raise ValueError("train_step: minibatch arrays of different lengths")
        if x.shape[0] > B * int(nmax):
            raise ValueError(f"{x.shape[0]} nodes in {B} graphs exceed nmax {int(nmax)}")
        loss = bufs["loss"] if loss_out is None else loss_out
        t = GcnTrainDesc()
        t.B, t.NMAX, t.EMAX, t.mode, t.gamma = B, int(nmax), max(int(emax), 1), int(mode), float(gamma)
        t.x, t.node_ptr, t.esrc, t.edst, t.edge_ptr = x.data_ptr(), node_ptr.data_ptr(), esrc.data_ptr(), edst.data_ptr(), edge_ptr.data_ptr()
        t.q_other, t.action, t.reward, t.nonfinal = q_other.data_ptr(), action.data_ptr(), reward.data_ptr(), nonfinal.data_ptr()
        t.workspace, t.partial, t.grad, t.loss = bufs["ws"].data_ptr(), bufs["partial"].data_ptr(), bufs["grad"].data_ptr(), loss.data_ptr()
        t.out = bufs["out"].data_ptr() if want_out else None
        t.layout = self.layout
        _lib.check(self.lib.mdq_gcn_train_step(C.byref(d), C.byref(t), _lib.stream_ptr(stream)), "mdq_gcn_train_step")
        return (loss, bufs["grad"]) + ((bufs["out"],) if want_out else ())

    @torch.no_grad()
    def forward(self, data, return_embedding=False, stream=None):
        """`data`: Data / Batch with x (N,F), edge_index (2,E) global node ids, batch (N,) or None."""
        self._pack(stream)
        d = self.desc
        dev = self.net.lin1.weight.device
        x = data.x.to(dev, torch.float32)
        from .airfoilgcnn import AirfoilGCNN
        if isinstance(self.net, AirfoilGCNN):
            x = x[:, [2, 3]]
        x = x.contiguous()
        n = x.shape[0]
        batch = data.batch if getattr(data, "batch", None) is not None else torch.zeros(n, dtype=torch.long, device=dev)
        batch = batch.to(dev)
        B = int(batch.max().item()) + 1
        counts = torch.bincount(batch, minlength=B)
        node_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        node_ptr[1:] = torch.cumsum(counts, 0)
        ei = data.edge_index.to(dev).reshape(2, -1)
        eg = batch[ei[0]] if ei.numel() else torch.zeros(0, dtype=torch.long, device=dev)
        order = torch.argsort(eg, stable=True)
        ei, eg = ei[:, order], eg[order]
        ecounts = torch.bincount(eg, minlength=B)
        edge_ptr = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        edge_ptr[1:] = torch.cumsum(ecounts, 0)
        esrc = (ei[0] - node_ptr[eg]).to(torch.int32).contiguous()
        edst = (ei[1] - node_ptr[eg]).to(torch.int32).contiguous()
        NMAX = int(counts.max().item())
        EMAX = max(int(ecounts.max().item()) if ecounts.numel() else 0, 1)
        node_ptr32, edge_ptr32 = node_ptr.to(torch.int32), edge_ptr.to(torch.int32)
        emb = torch.empty((B, 2 * d.C), dtype=torch.float32, device=dev)
        out = torch.empty((B, d.out_dim), dtype=torch.float32, device=dev)
        rc = self.lib.mdq_gcn_forward(C.byref(d), B, NMAX, EMAX, x.data_ptr(), node_ptr32.data_ptr(),
                                      esrc.data_ptr(), edst.data_ptr(), edge_ptr32.data_ptr(), emb.data_ptr(),
                                      out.data_ptr(), _lib.stream_ptr(stream))
        _lib.check(rc, "mdq_gcn_forward")
        return (out, emb) if return_embedding else out


    @torch.no_grad()
    def forward_arrays(self, x, node_ptr, esrc, edst, edge_ptr, nmax, emax, stream=None, edge_counts=None,
                       return_perm=False, return_status=False, pack=True, edge_cnt=None):
        """Same launch on pre-built arrays (what `VecEnv2DAirfoil.get_state` returns): x (sumN,F) f32,
        node_ptr / edge_ptr (B+1,) i32, esrc / edst (sumE,) i32 local node ids.  `nmax` / `emax` size the kernel's LDS
        carve-up: `edge_counts` (host array of the per-graph edge counts, where the caller has them) is checked against
        `emax` before the launch, and the kernel itself refuses larger graphs (NaN outputs, `return_status`).
        `return_perm`: also the (B, levels, nmax) TopKPooling `perm` arrays.  `pack=False`: use the packed parameter copy
        as it is (the caller has called `_pack()` at a point ordered against the writers of the parameters).  `edge_cnt`
        ((B,) i32 device tensor): the edge lists are PADDED - esrc / edst are (B, emax) arrays, graph b owns the first
        edge_cnt[b] slots of its row (what `mdq_env_topology` writes) - and `edge_ptr` is ignored."""
        if pack:
            self._pack(stream)
        elif self.desc is None:
            raise _lib.MeshDQNHipError("forward_arrays(pack=False) before the first _pack()")
        d = self.desc
        B = node_ptr.numel() - 1
        if edge_counts is not None and len(edge_counts) and int(max(edge_counts)) > int(emax):
            raise ValueError(f"graph with {int(max(edge_counts))} edges exceeds emax {int(emax)}")
        x = x.reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
        if x.shape[0] > B * int(nmax):
            raise ValueError(f"{x.shape[0]} nodes in {B} graphs exceed nmax {int(nmax)}")
        emb = torch.empty((B, 2 * d.C), dtype=torch.float32, device=x.device)
        out = torch.empty((B, d.out_dim), dtype=torch.float32, device=x.device)
        perm = torch.full((B, d.nlevels, int(nmax)), -1, dtype=torch.int32, device=x.device) if return_perm else None
        status = torch.zeros(B, dtype=torch.int32, device=x.device) if return_status else None
        if edge_cnt is not None:
            if esrc.shape[-1] != int(emax) or esrc.numel() != B * int(emax) or edge_cnt.numel() != B:
                raise ValueError("padded edge lists must be (B, emax) with (B,) counts")
            rc = self.lib.mdq_gcn_forward_padded(C.byref(d), B, int(nmax), int(emax), x.data_ptr(), node_ptr.data_ptr(),
                                                 esrc.data_ptr(), edst.data_ptr(), edge_cnt.data_ptr(), emb.data_ptr(),
                                                 out.data_ptr(), None if perm is None else perm.data_ptr(),
                                                 None if status is None else status.data_ptr(), _lib.stream_ptr(stream))
            _lib.check(rc, "mdq_gcn_forward_padded")
        else:
            rc = self.lib.mdq_gcn_forward_ex(C.byref(d), B, int(nmax), max(int(emax), 1), x.data_ptr(), node_ptr.data_ptr(),
                                             esrc.data_ptr(), edst.data_ptr(), edge_ptr.data_ptr(), emb.data_ptr(),
                                             out.data_ptr(), None if perm is None else perm.data_ptr(),
                                             None if status is None else status.data_ptr(), _lib.stream_ptr(stream))
            _lib.check(rc, "mdq_gcn_forward_ex")
        if return_perm or return_status:
            return (out,) + ((perm,) if return_perm else ()) + ((status,) if return_status else ())
        return out


def node_removal_forward(net, data):
    if not hasattr(net, "_fused"):
        net._fused = FusedGcn(net)
    return net._fused.forward(data)
