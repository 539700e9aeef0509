"""CPU tests: product nn.Modules (vectorised torch) against the oracle (plain-loop restatement of the PyG
layer semantics) and hand-computed tiny-graph vectors."""
import math

import numpy as np
import pytest
import torch

from meshdqn_amd import airfoilgcnn as prod
from meshdqn_amd.data import Batch, Data, DataLoader
from oracle import gcn as ora


def _graph(rng, n, e, f):
    x = torch.from_numpy(rng.standard_normal((n, f))).float()
    ei = torch.from_numpy(rng.integers(0, n, size=(2, e))).long()
    return Data(x=x, edge_index=ei)


def test_tiny_graph_hand_vectors():
    """3-node path graph 0->1, 2->1, 1->0 with unit features: values checkable by hand."""
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    ei = torch.tensor([[0, 2, 1], [1, 1, 0]])
    sage = prod.SAGEConv(2, 1)
    with torch.no_grad():
        sage.lin_l.weight.copy_(torch.tensor([[1.0, 1.0]]))
        sage.lin_l.bias.fill_(0.5)
        sage.lin_r.weight.copy_(torch.tensor([[1.0, -1.0]]))
    # node0: mean of {x1} = (3,4) -> 7 + .5 + (1-2) = 6.5 ; node1: mean{x0,x2} = (3,4) -> 7.5 + (3-4) = 6.5
    # node2: no incoming edge -> 0 + .5 + (5-6) = -0.5
    assert torch.allclose(sage(x, ei).flatten(), torch.tensor([6.5, 6.5, -0.5]))
    gcn = prod.GCNConv(2, 1)
    with torch.no_grad():
        gcn.lin.weight.copy_(torch.tensor([[1.0, 0.0]]))
        gcn.bias.fill_(1.0)
    # h = (1,3,5); deg (incl. self loop) = (2,3,1)
    d = torch.tensor([2.0, 3.0, 1.0]).pow(-0.5)
    exp = torch.tensor([1 * d[0] * d[0] + 3 * d[1] * d[0], 3 * d[1] * d[1] + 1 * d[0] * d[1] + 5 * d[2] * d[1], 5 * d[2] * d[2]]) + 1.0
    assert torch.allclose(gcn(x, ei).flatten(), exp)
    pool = prod.TopKPooling(2, ratio=0.5)
    with torch.no_grad():
        pool.weight.copy_(torch.tensor([[3.0, 4.0]]))
    xo, eo, _, bo, perm, sc = pool(x, ei, None, None)
    s = torch.tanh((x @ torch.tensor([3.0, 4.0])) / 5.0)
    assert perm.tolist() == [2, 1]  # k = ceil(1.5) = 2, highest scores first
    assert torch.allclose(xo, x[[2, 1]] * s[[2, 1]].unsqueeze(1))
    assert eo.tolist() == [[0], [1]]  # only edge 2->1 survives, relabelled
    assert bo.tolist() == [0, 0]


@pytest.mark.parametrize("cls,kw,feat", [("NodeRemovalNet", dict(output_dim=181, conv_width=128, topk=0.1), 17),
                                          ("AirfoilGCNN", dict(conv_width=64), 17)])
def test_networks_match_oracle(cls, kw, feat):
    rng = np.random.default_rng(5)
    net_p = getattr(prod, cls)(**kw)
    net_o = getattr(ora, cls)(**kw)
    if cls == "NodeRemovalNet":
        net_p.set_num_nodes(feat)
        net_o.set_num_nodes(feat)
    sd = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)) * 0.3).float() for k, v in net_p.state_dict().items()}
    net_p.load_state_dict(sd)
    net_o.load_state_dict(sd)  # identical PyG-style keys
    assert list(net_p.state_dict().keys()) == list(net_o.state_dict().keys())
    graphs = [_graph(rng, 180, 372, feat), _graph(rng, 180, 495, feat), _graph(rng, 37, 60, feat)]
    batch = Batch.from_data_list(graphs)
    with torch.no_grad():
        yp = net_p(batch)
        yo = net_o(batch)
        # a bare Data (batch=None) must work like in the reference (TopKPooling substitutes zeros)
        y1 = net_p(graphs[0])
    assert yp.shape == yo.shape == (3, kw.get("output_dim", 1))
    assert torch.allclose(yp, yo, rtol=1e-4, atol=1e-6)
    assert torch.allclose(y1[0], yp[0], rtol=1e-4, atol=1e-6)
    if cls == "NodeRemovalNet":
        assert torch.allclose(yp.sum(dim=1), torch.ones(3), atol=1e-5)  # softmax output
        n_par = sum(p.numel() for p in net_p.parameters())
        assert n_par == 173493
        with torch.no_grad():
            emb = net_p(batch, embedding=True)
        assert emb.shape == (3, 256)


def test_weight_accessors_and_loader():
    net = prod.NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    rng = np.random.default_rng(0)
    graphs = [_graph(rng, 180, 400, 17) for _ in range(5)]
    loader = DataLoader(graphs, batch_size=2)
    assert len(loader) == 3
    out = [net(b) for b in loader]
    assert [o.shape[0] for o in out] == [2, 2, 1]
    loss = sum(o[:, 3].sum() for o in out)
    loss.backward()
    grads = net.get_gradients()
    w = net.get_weights()
    net2 = prod.NodeRemovalNet(181, conv_width=128, topk=0.1)
    net2.set_num_nodes(17)
    net2.set_weights(w)
    net2.set_gradients(grads)
    flat = net2.flat_gradients()
    assert flat.numel() == 173493
    assert torch.allclose(flat, net.flat_gradients())
    net2.set_flat_gradients(flat * 2)
    assert torch.allclose(net2.flat_gradients(), flat * 2)


def test_dense_static_shape_path_equals_ragged_path():
    """`forward_dense` (fixed shapes, masked padded edges: the HIP-graph-capturable training path) against `forward`
    on the equivalent Batch: values and parameter gradients."""
    from meshdqn_amd.airfoilgcnn import dense_batch
    rng = np.random.default_rng(9)
    net = prod.NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    sd = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)) * 0.3).float() for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    graphs = [_graph(rng, 180, e, 17) for e in (372, 495, 0, 611, 37)]
    batch = Batch.from_data_list(graphs)
    x, src, dst, mask = dense_batch(graphs, 640)
    q_r = net(batch)
    q_d = net.forward_dense(x, src, dst, mask)
    assert torch.allclose(q_d, q_r, rtol=1e-4, atol=1e-6)
    assert torch.allclose(net.forward_dense(x, src, dst, mask, embedding=True), net(batch, embedding=True), rtol=1e-4, atol=1e-5)
    act = torch.tensor([[3], [50], [180], [7], [99]])
    for q in (q_r, q_d):
        net.zero_grad()
        torch.nn.functional.huber_loss(q.gather(1, act).squeeze(1), torch.linspace(0, 1, 5)).backward()
        if q is q_r:
            g_r = [p.grad.clone() if p.grad is not None else None for p in net.parameters()]
    for p, g in zip(net.parameters(), g_r):
        assert (p.grad is None) == (g is None)
        if g is not None:
            assert torch.allclose(p.grad, g, rtol=2e-3, atol=1e-7), (p.shape, (p.grad - g).abs().max())
