"""Fused HIP inference of `NodeRemovalNet` / `AirfoilGCNN` (host side of `mdq_gcn_forward`).

Packs the module's parameters (transposed to [in][out] so that consecutive lanes read consecutive
output channels), builds the per-graph node / edge ranges of a `Batch` and launches the two
kernels of meshdqn_amd/csrc/mdq_gcn.hip.  Inference only (no autograd); training uses the
module's regular `forward`.
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


class FusedGcn:
    """Device-side packed copy of a network's parameters + launcher."""

    def __init__(self, net):
        self.net = net
        self.lib = _lib.load()
        self._version = None
        self._keep = []
        self.desc = None

    def _pack(self):
        net = self.net
        levels, softmax = _levels_of(net)
        version = tuple(p._version for p in net.parameters()) + (id(net.conv1),)
        if version == self._version:
            return
        dev = net.lin1.weight.device
        if dev.type != "cuda":
            raise _lib.MeshDQNHipError("fused GCN forward needs the module on a GPU (no CPU fallback)")
        keep = []

        def dv(t, transpose=False):
            t = t.detach().to(torch.float32)
            t = t.t().contiguous() if transpose else t.contiguous()
            keep.append(t)
            return t.data_ptr()

        d = GcnNet()
        d.nlevels = len(levels)
        d.C = net.lin1.weight.shape[1] // 2
        d.out_dim = net.lin3.weight.shape[0]
        d.ratio = float(levels[0][1].ratio)
        d.softmax = 1 if softmax else 0
        for l, (conv, pool) in enumerate(levels):
            lv = d.levels[l]
            if hasattr(conv, "lin_l"):
                lv.type = 0
                lv.fin = conv.lin_l.weight.shape[1]
                lv.w_l = dv(conv.lin_l.weight, True)
                lv.b = dv(conv.lin_l.bias)
                lv.w_r = dv(conv.lin_r.weight, True)
            else:
                lv.type = 1
                lv.fin = conv.lin.weight.shape[1]
                lv.w_l = dv(conv.lin.weight, True)
                lv.b = dv(conv.bias)
                lv.w_r = None
            lv.pool_w = dv(pool.weight.reshape(-1))
            if l > 0 and lv.fin != d.C:
                raise ValueError("inner levels must have conv_width inputs")
        d.fin0 = d.levels[0].fin
        d.lin1_w, d.lin1_b = dv(net.lin1.weight, True), dv(net.lin1.bias)
        d.lin2_w, d.lin2_b = dv(net.lin2.weight, True), dv(net.lin2.bias)
        d.lin3_w, d.lin3_b = dv(net.lin3.weight, True), dv(net.lin3.bias)
        if net.lin1.weight.shape[0] != 128 or net.lin2.weight.shape[0] != 64:
            raise ValueError("head must be 2C -> 128 -> 64 -> out (as in the reference)")
        self.desc, self._keep, self._version = d, keep, version

    @torch.no_grad()
    def forward(self, data, return_embedding=False, stream=None):
        """`data`: Data / Batch with x (N,F), edge_index (2,E) global node ids, batch (N,) or None."""
        self._pack()
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
                       return_perm=False, return_status=False):
        """Same launch on pre-built arrays (what `VecEnv2DAirfoil.get_state` returns): x (sumN,F) f32,
        node_ptr / edge_ptr (B+1,) i32, esrc / edst (sumE,) i32 local node ids.  `nmax` / `emax` size the kernel's LDS
        carve-up: `edge_counts` (host array of the per-graph edge counts, where the caller has them) is checked against
        `emax` before the launch, and the kernel itself refuses larger graphs (NaN outputs, `return_status`).
        `return_perm`: also the (B, levels, nmax) TopKPooling `perm` arrays."""
        self._pack()
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
