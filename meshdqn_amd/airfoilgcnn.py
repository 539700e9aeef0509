"""`NodeRemovalNet` / `AirfoilGCNN` - the reference's nn.Module surfaces (airfoilgcnn.py:24-209)
without torch_geometric.

Constructor signatures, attribute names and `state_dict` keys equal the reference's
(PyG < 2.3 names: conv*.lin_l/lin_r/lin, pool*.weight, lin*), so checkpoints written
by `airfoil_dqn.py:214-218` load unchanged.  Two execution paths:

  * autograd path (training, any device): vectorised PyTorch ops (index_add_ scatter,
    sort-based per-graph top-k);
  * fused inference path (`forward_fused`, MI355X): one HIP workgroup per graph runs the whole
    network out of LDS (meshdqn_amd/csrc/mdq_gcn.hip) - used by `select_action`.

Also defines the four accessors the reference's trainer calls but never defines
(`get_weights/set_weights/get_gradients/set_gradients`, airfoil_dqn.py:194-206,291-310) with the
semantics of the Ray parameter-server example they were taken from.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

# the reference picks cuda:0 when available only to place the re-created conv1 (airfoilgcnn.py:16-22,80)
device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")


class Linear(nn.Module):
    """PyG `Linear`: weight (out,in) ~ kaiming_uniform(a=sqrt(5)), optional zero bias."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class SAGEConv(nn.Module):
    """out_i = lin_l(mean_{j->i} x_j) + lin_r(x_i)   (aggr='mean', root_weight, bias in lin_l)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin_l = Linear(in_channels, out_channels, bias=True)
        self.lin_r = Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index):
        n = x.shape[0]
        src, dst = edge_index[0], edge_index[1]
        agg = torch.zeros_like(x).index_add_(0, dst, x[src])
        cnt = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(
            0, dst, torch.ones(dst.numel(), dtype=x.dtype, device=x.device))
        agg = agg / cnt.clamp(min=1).unsqueeze(1)
        return self.lin_l(agg) + self.lin_r(x)


class GCNConv(nn.Module):
    """out = D^-1/2 (A + I) D^-1/2 X W + b   (degree over targets incl. the self loop)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = Linear(in_channels, out_channels, bias=False)
        nn.init.xavier_uniform_(self.lin.weight)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index):
        n = x.shape[0]
        src, dst = edge_index[0], edge_index[1]
        deg = torch.ones(n, dtype=x.dtype, device=x.device).index_add_(
            0, dst, torch.ones(dst.numel(), dtype=x.dtype, device=x.device))
        dis = deg.pow(-0.5)
        h = self.lin(x)
        out = h * (dis * dis).unsqueeze(1)
        out = out.index_add(0, dst, h[src] * (dis[src] * dis[dst]).unsqueeze(1))
        return out + self.bias


class TopKPooling(nn.Module):
    """score = tanh(x.w/|w|); keep the ceil(ratio*n) highest per graph; x' = x[perm]*score[perm]."""

    def __init__(self, in_channels, ratio=0.5):
        super().__init__()
        self.in_channels, self.ratio = in_channels, ratio
        self.weight = nn.Parameter(torch.empty(1, in_channels))
        bound = 1.0 / math.sqrt(in_channels)
        nn.init.uniform_(self.weight, -bound, bound)

    def forward(self, x, edge_index, edge_attr=None, batch=None):
        n = x.shape[0]
        if batch is None:
            batch = torch.zeros(n, dtype=torch.long, device=x.device)
        score = torch.tanh((x * self.weight).sum(dim=-1) / self.weight.norm(p=2, dim=-1))
        B = int(batch.max().item()) + 1 if n else 0
        order = torch.argsort(score, descending=True, stable=True)
        order = order[torch.argsort(batch[order], stable=True)]  # grouped by graph, descending score inside
        counts = torch.bincount(batch, minlength=B)
        k = torch.ceil(self.ratio * counts.to(torch.float64)).to(torch.long)
        starts = torch.cumsum(counts, 0) - counts
        bs = batch[order]
        rank = torch.arange(n, device=x.device) - starts[bs]
        perm = order[rank < k[bs]]
        xo = x[perm] * score[perm].unsqueeze(-1)
        new_id = torch.full((n,), -1, dtype=torch.long, device=x.device)
        new_id[perm] = torch.arange(perm.numel(), device=x.device)
        if edge_index.numel():
            s, d = new_id[edge_index[0]], new_id[edge_index[1]]
            keep = (s >= 0) & (d >= 0)
            edge_index = torch.stack([s[keep], d[keep]])
        return xo, edge_index, None, batch[perm], perm, score[perm]


def global_max_pool(x, batch):
    B = int(batch.max().item()) + 1
    out = torch.full((B, x.shape[1]), -float("inf"), dtype=x.dtype, device=x.device)
    return out.scatter_reduce(0, batch.unsqueeze(1).expand_as(x), x, reduce="amax", include_self=True)


def global_mean_pool(x, batch):
    B = int(batch.max().item()) + 1
    s = torch.zeros((B, x.shape[1]), dtype=x.dtype, device=x.device).index_add_(0, batch, x)
    c = torch.bincount(batch, minlength=B).clamp(min=1).to(x.dtype)
    return s / c.unsqueeze(1)


gmp, gap = global_max_pool, global_mean_pool


# ---------------------------------------------------------------------------------------------------------------
# Static-shape ("dense batch") formulation of the same layers for training: every graph of a minibatch has the same
# number of nodes (N_closest rows, padded states are zero rows exactly as in the reference) and its edge list is
# padded to a fixed length with a 0/1 mask.  No host synchronisation, no data-dependent shapes: the whole
# forward + backward is a fixed sequence of kernels that a HIP graph can replay (trainer.DQNTrainer).  Numerically
# the same sums as the ragged path above, in a different order.

def _dense_adjacency(n: int, src, dst, m):
    """A[b, i, j] = number of edges j -> i of graph b (duplicates counted, padded edges have weight 0): with the graphs
    of a minibatch padded to n nodes, every aggregation of the conv / pool stack is a small batched matrix product -
    no gathers or atomic scatters over the e_max edge slots at any level, in the forward or the backward pass."""
    B = src.shape[0]
    A = torch.zeros((B, n * n), dtype=m.dtype, device=m.device).scatter_add(1, dst * n + src, m)
    return A.view(B, n, n)


def _dense_sage(conv: SAGEConv, x, A):
    cnt = A.sum(dim=-1, keepdim=True)
    agg = torch.bmm(A, x) / cnt.clamp(min=1)
    return conv.lin_l(agg) + conv.lin_r(x)


def _dense_gcn(conv: GCNConv, x, A):
    dis = (1.0 + A.sum(dim=-1, keepdim=True)).pow(-0.5)          # degree over targets incl. the self-loop
    h = conv.lin(x) * dis
    return (torch.bmm(A, h) + h) * dis + conv.bias


def _dense_topk(pool: TopKPooling, x, A):
    B, n, C = x.shape
    k = int(math.ceil(pool.ratio * n))
    score = torch.tanh((x * pool.weight).sum(dim=-1) / pool.weight.norm(p=2, dim=-1))
    # stable descending order (ties keep the lower node id first, like the ragged path's stable argsort)
    order = torch.argsort(score, dim=1, descending=True, stable=True)[:, :k]
    sc = torch.gather(score, 1, order)
    xo = torch.gather(x, 1, order.unsqueeze(-1).expand(-1, -1, C)) * sc.unsqueeze(-1)
    # induced subgraph on the kept nodes, relabelled in score order
    A2 = torch.gather(torch.gather(A, 1, order.unsqueeze(-1).expand(-1, -1, n)), 2, order.unsqueeze(1).expand(-1, k, -1))
    return xo, A2


def _dense_readout(x):
    return torch.cat([x.max(dim=1).values, x.mean(dim=1)], dim=1)


class _WeightAccessors:
    """`get_weights/set_weights/get_gradients/set_gradients` (called at airfoil_dqn.py:194-206,
    291-310, never defined by the reference): Ray parameter-server example semantics."""

    def get_weights(self):
        return {k: v.detach().cpu() for k, v in self.state_dict().items()}

    def set_weights(self, weights):
        self.load_state_dict(weights)

    def get_gradients(self):
        return [None if p.grad is None else p.grad.detach().cpu().numpy() for p in self.parameters()]

    def set_gradients(self, gradients):
        for g, p in zip(gradients, self.parameters()):
            if g is not None:
                p.grad = torch.from_numpy(np.asarray(g)).to(p.device, p.dtype)

    # flat views for the RCCL all-reduce of the data-parallel trainer
    def flat_gradients(self) -> torch.Tensor:
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.parameters()])

    def unused_parameters(self):
        """Parameters that are constructed (and part of the `state_dict`) but never enter `forward` - the reference's
        conv3 / pool3 / conv6 / pool6 of `NodeRemovalNet` (airfoilgcnn.py:106-110,124-128).  Their `.grad` stays None,
        as it does under autograd in the reference, so that an optimiser with weight decay leaves them alone."""
        return []

    def set_flat_gradients(self, flat: torch.Tensor):
        """Gradients from a flat buffer laid out like `flat_gradients()` (all parameters, so that the message of the
        all-reduce has the same 173 493 elements on every rank); entries of `unused_parameters()` are ignored."""
        skip = {id(p) for p in self.unused_parameters()}
        off = 0
        for p in self.parameters():
            n = p.numel()
            p.grad = None if id(p) in skip else flat[off:off + n].view_as(p).clone()
            off += n


class NodeRemovalNet(nn.Module, _WeightAccessors):
    def __init__(self, output_dim, conv_width=64, topk=0.5, initial_num_nodes=None):
        super(NodeRemovalNet, self).__init__()
        self.conv_width = conv_width
        self.topk = topk
        self.output_dim = output_dim
        self.initial_num_nodes = initial_num_nodes
        self.conv1 = SAGEConv(2, conv_width)
        self.pool1 = TopKPooling(conv_width, ratio=topk)
        self.conv2 = SAGEConv(conv_width, conv_width)
        self.pool2 = TopKPooling(conv_width, ratio=topk)
        self.conv3 = SAGEConv(conv_width, conv_width)
        self.pool3 = TopKPooling(conv_width, ratio=topk)
        self.conv4 = GCNConv(conv_width, conv_width)
        self.pool4 = TopKPooling(conv_width, ratio=topk)
        self.conv5 = GCNConv(conv_width, conv_width)
        self.pool5 = TopKPooling(conv_width, ratio=topk)
        self.conv6 = GCNConv(conv_width, conv_width)
        self.pool6 = TopKPooling(conv_width, ratio=topk)
        self.lin1 = torch.nn.Linear(2 * conv_width, 128)
        self.lin2 = torch.nn.Linear(128, 64)
        self.lin3 = torch.nn.Linear(64, output_dim)
        torch.manual_seed(0)
        self.reset()

    def unused_parameters(self):
        return [p for m in (self.conv3, self.pool3, self.conv6, self.pool6) for p in m.parameters()]

    def reset(self):
        """Initialisation scheme of airfoilgcnn.py:50-76."""
        for conv in (self.conv1, self.conv2, self.conv3):
            nn.init.xavier_normal_(conv.lin_l.weight, gain=0.9)
            nn.init.normal_(conv.lin_l.bias)
            nn.init.xavier_normal_(conv.lin_r.weight, gain=0.9)
        for conv in (self.conv4, self.conv5, self.conv6):
            nn.init.xavier_normal_(conv.lin.weight, gain=0.9)
        for lin in (self.lin1, self.lin2, self.lin3):
            nn.init.xavier_normal_(lin.weight, gain=0.9)
            nn.init.normal_(lin.bias)

    def set_num_nodes(self, initial_num_nodes):
        """Re-creates conv1 with `initial_num_nodes` input features and default init (airfoilgcnn.py:78-80)."""
        self.initial_num_nodes = initial_num_nodes
        self.conv1 = SAGEConv(self.initial_num_nodes, self.conv_width).to(self.lin1.weight.device)

    def set_removable(self, removable):
        self.removable = removable

    def forward(self, data, embedding=False):
        x, edge_index, batch = data.x.float(), data.edge_index, data.batch
        x = F.relu(self.conv1(x, edge_index))
        x, edge_index, _, batch, _, _ = self.pool1(x, edge_index, None, batch)
        x1 = torch.cat([gmp(x, batch), gap(x, batch)], dim=1)
        x = F.relu(self.conv2(x, edge_index))
        x, edge_index, _, batch, _, _ = self.pool2(x, edge_index, None, batch)
        x2 = torch.cat([gmp(x, batch), gap(x, batch)], dim=1)
        # conv3/pool3 and conv6/pool6 exist (state_dict) but are skipped (airfoilgcnn.py:106-110,124-128)
        x = F.relu(self.conv4(x, edge_index))
        x, edge_index, _, batch, _, _ = self.pool4(x, edge_index, None, batch)
        x4 = torch.cat([gmp(x, batch), gap(x, batch)], dim=1)
        x = F.relu(self.conv5(x, edge_index))
        x, edge_index, _, batch, _, _ = self.pool5(x, edge_index, None, batch)
        x5 = torch.cat([gmp(x, batch), gap(x, batch)], dim=1)
        x = x1 + x2 + x4 + x5
        if embedding:
            return x
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=0.0, training=self.training)
        x = F.relu(self.lin2(x))
        x = self.lin3(x)
        x = F.softmax(x, dim=1)  # pick which vertex to remove
        return x

    def forward_fused(self, data):
        """Inference (no autograd) on the fused HIP kernels: one workgroup per graph, MFMA head."""
        from .gcn_fused import node_removal_forward
        return node_removal_forward(self, data)

    def forward_dense(self, x, src, dst, mask, embedding=False):
        """Autograd path with static shapes: x (B,n,F) f32, src/dst (B,E) int64 local node ids, mask (B,E) 0/1 float
        (padded edges have mask 0 and any valid node id).  Same function as `forward` on the equivalent Batch."""
        A = _dense_adjacency(x.shape[1], src, dst, mask)
        x = F.relu(_dense_sage(self.conv1, x.float(), A))
        x, A = _dense_topk(self.pool1, x, A)
        x1 = _dense_readout(x)
        x = F.relu(_dense_sage(self.conv2, x, A))
        x, A = _dense_topk(self.pool2, x, A)
        x2 = _dense_readout(x)
        x = F.relu(_dense_gcn(self.conv4, x, A))
        x, A = _dense_topk(self.pool4, x, A)
        x4 = _dense_readout(x)
        x = F.relu(_dense_gcn(self.conv5, x, A))
        x, A = _dense_topk(self.pool5, x, A)
        x5 = _dense_readout(x)
        x = x1 + x2 + x4 + x5
        if embedding:
            return x
        x = F.relu(self.lin1(x))
        x = F.relu(self.lin2(x))
        return F.softmax(self.lin3(x), dim=1)


def dense_batch(data_list, e_max: int, device=None):
    """Stack graphs with equal node counts into the static-shape inputs of `forward_dense`:
    x (B,n,F), src / dst (B,e_max) int64, mask (B,e_max) float32."""
    B = len(data_list)
    n = data_list[0].x.shape[0]
    dev = device if device is not None else data_list[0].x.device
    x = torch.stack([d.x.to(dev) for d in data_list]).float()
    src = torch.zeros((B, e_max), dtype=torch.long, device=dev)
    dst = torch.zeros((B, e_max), dtype=torch.long, device=dev)
    mask = torch.zeros((B, e_max), dtype=torch.float32, device=dev)
    cnt = [int(d.edge_index.shape[1]) for d in data_list]
    if any(d.x.shape[0] != n for d in data_list):
        raise ValueError("dense_batch needs graphs with equal node counts")
    if max(cnt) > e_max:
        raise ValueError(f"graph with {max(cnt)} edges exceeds e_max {e_max}")
    total = sum(cnt)
    if total:
        # one scatter for the whole minibatch (row b, column = position inside graph b's edge list)
        # (index arithmetic in numpy, not torch-CPU: torch's intra-op thread pool on a many-core host keeps spinning
        # after a parallel region and starves the autograd thread; measured 25 ms per backward on a 256-CPU box)
        c = np.asarray(cnt, dtype=np.int64)
        rows = np.repeat(np.arange(B, dtype=np.int64), c)
        cols = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(c) - c, c)
        lin = torch.from_numpy(rows * e_max + cols).to(dev)   # scatter_ on the flat views below
        allE = torch.cat([d.edge_index.to(dev) for d in data_list], dim=1)
        src.view(-1).scatter_(0, lin, allE[0])
        dst.view(-1).scatter_(0, lin, allE[1])
        mask.view(-1).scatter_(0, lin, torch.ones(total, dtype=torch.float32, device=dev))
    return x, src, dst, mask


class AirfoilGCNN(nn.Module, _WeightAccessors):
    def __init__(self, conv_width=64):
        super(AirfoilGCNN, self).__init__()
        topk = 0.5
        self.conv1 = SAGEConv(2, conv_width)
        self.pool1 = TopKPooling(conv_width, ratio=topk)
        self.conv2 = SAGEConv(conv_width, conv_width)
        self.pool2 = TopKPooling(conv_width, ratio=topk)
        self.conv3 = SAGEConv(conv_width, conv_width)
        self.pool3 = TopKPooling(conv_width, ratio=topk)
        self.conv4 = GCNConv(conv_width, conv_width)
        self.pool4 = TopKPooling(conv_width, ratio=topk)
        self.conv5 = GCNConv(conv_width, conv_width)
        self.pool5 = TopKPooling(conv_width, ratio=topk)
        self.conv6 = GCNConv(conv_width, conv_width)
        self.pool6 = TopKPooling(conv_width, ratio=topk)
        self.lin1 = torch.nn.Linear(2 * conv_width, 128)
        self.lin2 = torch.nn.Linear(128, 64)
        self.lin3 = torch.nn.Linear(64, 1)

    def forward(self, data):
        x, edge_index, batch = data.x.float(), data.edge_index, data.batch
        x = x[:, [2, 3]]
        outs = []
        for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv3, self.pool3),
                           (self.conv4, self.pool4), (self.conv5, self.pool5), (self.conv6, self.pool6)):
            x = F.relu(conv(x, edge_index))
            x, edge_index, _, batch, _, _ = pool(x, edge_index, None, batch)
            outs.append(torch.cat([gmp(x, batch), gap(x, batch)], dim=1))
        x = outs[0] + outs[1] + outs[2] + outs[3] + outs[4] + outs[5]
        x = F.relu(self.lin1(x))
        x = F.dropout(x, p=0.0, training=self.training)
        x = F.relu(self.lin2(x))
        x = self.lin3(x)
        return x

    def forward_fused(self, data):
        """Inference (no autograd) on the fused HIP kernels."""
        from .gcn_fused import node_removal_forward
        return node_removal_forward(self, data)
