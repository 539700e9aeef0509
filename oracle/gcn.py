"""Oracle: the reference's Q-networks without torch_geometric (plain PyTorch, CPU, fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the
reference holds no test or golden vector for the networks and torch_geometric
cannot be imported here; the PyG layer semantics below are restated from their
documented behaviour at the call sites `airfoilgcnn.py:30-44,94-134`
(SURVEY.md Appendix A.5).

  SAGEConv(in,out)   out_i = W_l mean_{j->i} x_j + b_l + W_r x_i        (lin_l.weight/bias, lin_r.weight)
  GCNConv(in,out)    out_i = sum_{j->i or j=i} d_j^-1/2 d_i^-1/2 (W x_j) + b,  d = in-degree incl. self loop
  TopKPooling(C,r)   score = tanh(x.w/|w|); keep ceil(r n) best per graph; x' = x[perm] * score[perm]
  global_max_pool / global_mean_pool per graph
`state_dict` keys are the PyG (<2.3) names: conv{1,2,3}.lin_l.{weight,bias}, conv{1,2,3}.lin_r.weight,
conv{4,5,6}.{lin.weight,bias}, pool{1..6}.weight, lin{1,2,3}.{weight,bias}.
"""
import math

import torch
from torch import nn
import torch.nn.functional as F


class _Lin(nn.Module):
    def __init__(self, i, o, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i).uniform_(-1 / math.sqrt(i), 1 / math.sqrt(i)))
        if bias:
            self.bias = nn.Parameter(torch.zeros(o))
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class SAGEConv(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.lin_l = _Lin(i, o, True)
        self.lin_r = _Lin(i, o, False)

    def forward(self, x, edge_index):
        n = x.shape[0]
        src, dst = edge_index[0], edge_index[1]
        agg = torch.zeros((n, x.shape[1]), dtype=x.dtype)
        cnt = torch.zeros(n, dtype=x.dtype)
        for e in range(src.numel()):  # plain loop: independent of any scatter implementation
            agg[dst[e]] += x[src[e]]
            cnt[dst[e]] += 1
        agg = agg / cnt.clamp(min=1).unsqueeze(1)
        return self.lin_l(agg) + self.lin_r(x)


class GCNConv(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.lin = _Lin(i, o, False)
        self.bias = nn.Parameter(torch.zeros(o))

    def forward(self, x, edge_index):
        n = x.shape[0]
        src, dst = edge_index[0], edge_index[1]
        deg = torch.ones(n, dtype=x.dtype)  # self loop
        for e in range(src.numel()):
            deg[dst[e]] += 1
        dis = deg.pow(-0.5)
        h = self.lin(x)
        out = h * (dis * dis).unsqueeze(1)  # self loops
        for e in range(src.numel()):
            out[dst[e]] = out[dst[e]] + dis[src[e]] * dis[dst[e]] * h[src[e]]
        return out + self.bias


class TopKPooling(nn.Module):
    def __init__(self, c, ratio):
        super().__init__()
        self.ratio = ratio
        self.weight = nn.Parameter(torch.empty(1, c).uniform_(-1 / math.sqrt(c), 1 / math.sqrt(c)))

    def forward(self, x, edge_index, edge_attr, batch):
        if batch is None:
            batch = torch.zeros(x.shape[0], dtype=torch.long)
        score = torch.tanh((x * self.weight).sum(dim=1) / self.weight.norm(p=2))
        perm = []
        for g in range(int(batch.max()) + 1 if batch.numel() else 0):
            idx = torch.nonzero(batch == g)[:, 0]
            k = int(math.ceil(self.ratio * idx.numel()))
            order = torch.argsort(score[idx], descending=True, stable=True)[:k]
            perm.append(idx[order])
        perm = torch.cat(perm) if perm else torch.zeros(0, dtype=torch.long)
        xo = x[perm] * score[perm].unsqueeze(1)
        new_id = torch.full((x.shape[0],), -1, dtype=torch.long)
        new_id[perm] = torch.arange(perm.numel())
        if edge_index.numel():
            s, d = new_id[edge_index[0]], new_id[edge_index[1]]
            keep = (s >= 0) & (d >= 0)
            ei = torch.stack([s[keep], d[keep]])
        else:
            ei = edge_index
        return xo, ei, None, batch[perm], perm, score[perm]


def gmp(x, batch):
    B = int(batch.max()) + 1
    return torch.stack([x[batch == g].max(dim=0)[0] for g in range(B)])


def gap(x, batch):
    B = int(batch.max()) + 1
    return torch.stack([x[batch == g].mean(dim=0) for g in range(B)])


class NodeRemovalNet(nn.Module):
    """airfoilgcnn.py:24-145."""

    def __init__(self, output_dim, conv_width=64, topk=0.5, initial_num_nodes=None):
        super().__init__()
        self.conv_width = conv_width
        self.conv1 = SAGEConv(2, conv_width)
        self.pool1 = TopKPooling(conv_width, topk)
        self.conv2 = SAGEConv(conv_width, conv_width)
        self.pool2 = TopKPooling(conv_width, topk)
        self.conv3 = SAGEConv(conv_width, conv_width)
        self.pool3 = TopKPooling(conv_width, topk)
        self.conv4 = GCNConv(conv_width, conv_width)
        self.pool4 = TopKPooling(conv_width, topk)
        self.conv5 = GCNConv(conv_width, conv_width)
        self.pool5 = TopKPooling(conv_width, topk)
        self.conv6 = GCNConv(conv_width, conv_width)
        self.pool6 = TopKPooling(conv_width, topk)
        self.lin1 = nn.Linear(2 * conv_width, 128)
        self.lin2 = nn.Linear(128, 64)
        self.lin3 = nn.Linear(64, output_dim)

    def set_num_nodes(self, n):
        self.initial_num_nodes = n
        self.conv1 = SAGEConv(n, self.conv_width)

    def forward(self, data, embedding=False, return_perm=False):
        """`return_perm`: also the four TopKPooling `perm` index arrays (airfoilgcnn.py:96,102,114,120; indices into the
        node set entering the level) and the pooling scores of the kept nodes - the index work of the Q-path."""
        x, edge_index, batch = data.x.float(), data.edge_index, data.batch
        perms, scores = [], []
        x = F.relu(self.conv1(x, edge_index))
        x, edge_index, _, batch, perm, score = self.pool1(x, edge_index, None, batch)
        perms.append(perm), scores.append(score)
        x1 = torch.cat([gmp(x, batch), gap(x, batch)], dim=1)
        x = F.relu(self.conv2(x, edge_index))
        x, edge_index, _, batch, perm, score = self.pool2(x, edge_index, None, batch)
        perms.append(perm), scores.append(score)
        x2 = torch.cat([gmp(x, batch), gap(x, batch)], dim=1)
        x = F.relu(self.conv4(x, edge_index))
        x, edge_index, _, batch, perm, score = self.pool4(x, edge_index, None, batch)
        perms.append(perm), scores.append(score)
        x4 = torch.cat([gmp(x, batch), gap(x, batch)], dim=1)
        x = F.relu(self.conv5(x, edge_index))
        x, edge_index, _, batch, perm, score = self.pool5(x, edge_index, None, batch)
        perms.append(perm), scores.append(score)
        x5 = torch.cat([gmp(x, batch), gap(x, batch)], dim=1)
        x = x1 + x2 + x4 + x5
        if not embedding:
            x = F.relu(self.lin1(x))
            x = F.relu(self.lin2(x))
            x = self.lin3(x)
            x = F.softmax(x, dim=1)
        return (x, perms, scores) if return_perm else x


def _node_removal_pool_scores(self, data):
    """TopKPooling scores of EVERY node entering each of the four levels (for the tie analysis of the parity tests)."""
    x, edge_index, batch = data.x.float(), data.edge_index, data.batch
    out = []
    for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv4, self.pool4), (self.conv5, self.pool5)):
        x = F.relu(conv(x, edge_index))
        out.append(torch.tanh((x * pool.weight).sum(dim=1) / pool.weight.norm(p=2)))
        x, edge_index, _, batch, _, _ = pool(x, edge_index, None, batch)
    return out


NodeRemovalNet.pool_scores = _node_removal_pool_scores


class AirfoilGCNN(nn.Module):
    """airfoilgcnn.py:148-209."""

    def __init__(self, conv_width=64):
        super().__init__()
        topk = 0.5
        self.conv1 = SAGEConv(2, conv_width)
        self.pool1 = TopKPooling(conv_width, topk)
        self.conv2 = SAGEConv(conv_width, conv_width)
        self.pool2 = TopKPooling(conv_width, topk)
        self.conv3 = SAGEConv(conv_width, conv_width)
        self.pool3 = TopKPooling(conv_width, topk)
        self.conv4 = GCNConv(conv_width, conv_width)
        self.pool4 = TopKPooling(conv_width, topk)
        self.conv5 = GCNConv(conv_width, conv_width)
        self.pool5 = TopKPooling(conv_width, topk)
        self.conv6 = GCNConv(conv_width, conv_width)
        self.pool6 = TopKPooling(conv_width, topk)
        self.lin1 = nn.Linear(2 * conv_width, 128)
        self.lin2 = nn.Linear(128, 64)
        self.lin3 = nn.Linear(64, 1)

    def forward(self, data):
        x, edge_index, batch = data.x.float(), data.edge_index, data.batch
        x = x[:, [2, 3]]
        outs = []
        for conv, pool in ((self.conv1, self.pool1), (self.conv2, self.pool2), (self.conv3, self.pool3),
                           (self.conv4, self.pool4), (self.conv5, self.pool5), (self.conv6, self.pool6)):
            x = F.relu(conv(x, edge_index))
            x, edge_index, _, batch, _, _ = pool(x, edge_index, None, batch)
            outs.append(torch.cat([gmp(x, batch), gap(x, batch)], dim=1))
        x = sum(outs)
        x = F.relu(self.lin1(x))
        x = F.relu(self.lin2(x))
        return self.lin3(x)
