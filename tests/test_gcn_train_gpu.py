"""GPU: the hand-written learning step of the graph Q-network (mdq_gcn_train_step: forward + double-DQN Huber loss +
backward without autograd, airfoil_dqn.py:240-310) against the oracle's plain-loop restatement (oracle/dqn.py over
oracle/gcn.py) and against torch autograd through the package's own ragged layers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _arrays(graphs, dev):
    """Concatenated arrays of `mdq_gcn_forward` / `mdq_gcn_train_step` from per-graph (x, edge_index) objects."""
    x = torch.cat([g.x for g in graphs]).float().to(dev).contiguous()
    node_ptr = torch.tensor(np.concatenate([[0], np.cumsum([g.x.shape[0] for g in graphs])]), dtype=torch.int32, device=dev)
    edge_ptr = torch.tensor(np.concatenate([[0], np.cumsum([g.edge_index.shape[1] for g in graphs])]), dtype=torch.int32, device=dev)
    esrc = torch.cat([g.edge_index[0] for g in graphs]).to(torch.int32).to(dev).contiguous()
    edst = torch.cat([g.edge_index[1] for g in graphs]).to(torch.int32).to(dev).contiguous()
    return x, node_ptr, esrc, edst, edge_ptr


def _minibatch(rng, B, n=180, f=17, out=181, emin=200, emax=500):
    from meshdqn_amd.data import Data

    def graph():
        e = int(rng.integers(emin, emax))
        return Data(x=torch.from_numpy(rng.standard_normal((n, f))).float(),
                    edge_index=torch.from_numpy(rng.integers(0, n, size=(2, e))).long())
    return [(graph(), int(rng.integers(0, out)), None if i % 3 == 2 else graph(), float(rng.uniform(-1, 1))) for i in range(B)]


def _fused_step(net, other, mb, select, gamma, dev, nmax, emax):
    """The learning step of `net` (the selected network) on the kernels; returns loss, {name: grad}."""
    from meshdqn_amd.gcn_fused import FusedGcn
    states = [s for s, _, _, _ in mb]
    nexts = [(n if n is not None else s) for s, _, n, _ in mb]      # terminal: own state as a masked placeholder
    action = torch.tensor([a for _, a, _, _ in mb], dtype=torch.int64, device=dev)
    reward = torch.tensor([r for _, _, _, r in mb], dtype=torch.float32, device=dev)
    nonfinal = torch.tensor([0.0 if n is None else 1.0 for _, _, n, _ in mb], dtype=torch.float32, device=dev)
    f_net, f_other = FusedGcn(net), FusedGcn(other)
    mine, theirs = (states, nexts) if select else (nexts, states)
    qo = f_other.forward_arrays(*_arrays(theirs, dev), nmax, emax)
    loss, flat = f_net.train_step(*_arrays(mine, dev), nmax, emax, 0 if select else 1, qo, action, reward, nonfinal, gamma)
    torch.cuda.synchronize()
    grads, off = {}, 0
    for k, p in net.named_parameters():
        grads[k] = flat[off:off + p.numel()].view_as(p).clone()
        off += p.numel()
    assert off == flat.numel()
    return float(loss.item()), grads, (f_net, qo, action, reward, nonfinal, mine)


@pytest.mark.parametrize("select", [True, False])
def test_fused_learning_step_matches_the_oracle_learning_step(lib_built, select):
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from oracle import gcn as ora
    from oracle.dqn import compute_gradients
    dev = torch.device("cuda")
    rng = np.random.default_rng(19 + select)
    nets, oras = [], []
    for _ in range(2):
        net = NodeRemovalNet(181, conv_width=128, topk=0.1)
        net.set_num_nodes(17)
        sd = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)) * 0.06).float() for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        o = ora.NodeRemovalNet(181, conv_width=128, topk=0.1)
        o.set_num_nodes(17)
        o.load_state_dict(sd)
        nets.append(net.to(dev))
        oras.append(o)
    mb = _minibatch(rng, 8)
    gamma = 0.9
    loss_o, grads_o = compute_gradients(oras[0], oras[1], mb, select, gamma)
    k = 0 if select else 1
    loss, grads, _ = _fused_step(nets[k], nets[1 - k], mb, select, gamma, dev, 180, 512)
    assert abs(loss - loss_o) < 1e-4 * max(abs(loss_o), 1e-3)
    scale = max(float(g.abs().max()) for g in grads_o.values() if g is not None)
    assert scale > 1e-6
    for name, g in grads_o.items():
        if g is None:     # conv3 / pool3 / conv6 / pool6: never in the forward pass, their slots stay zero
            assert float(grads[name].abs().max()) == 0.0, name
        else:
            assert float((grads[name].cpu() - g).abs().max()) < 1e-4 * scale, name


@pytest.mark.parametrize("kind", ["node_removal", "six_levels"])
def test_fused_learning_step_matches_autograd_and_is_reproducible(lib_built, kind):
    """Same loss and gradient as torch autograd through the package's ragged layers (index_add_ scatter, sort-based
    top-k), parameter by parameter, for the reference's NodeRemovalNet and for a six-level SAGE / GCN stack with
    TopK ratio 0.5 (wide levels: every backward branch with many kept rows and edges); bitwise equal on a second run."""
    from meshdqn_amd.airfoilgcnn import AirfoilGCNN, NodeRemovalNet
    from meshdqn_amd.data import Batch
    dev = torch.device("cuda")
    rng = np.random.default_rng(5)
    torch.manual_seed(3)
    if kind == "node_removal":
        mk = lambda: NodeRemovalNet(181, conv_width=128, topk=0.1)   # noqa: E731
        n, f, out, B = 180, 17, 181, 8
        nets = [mk(), mk()]
        for m in nets:
            m.set_num_nodes(17)
    else:
        n, f, out, B = 120, 2, 1, 6
        nets = [AirfoilGCNN(conv_width=64), AirfoilGCNN(conv_width=64)]
    for m in nets:
        m.to(dev)
    mb = _minibatch(rng, B, n=n, f=f, out=out, emin=150, emax=400)
    gamma = 0.8
    for select in (True, False):
        k = 0 if select else 1
        net, other = nets[k], nets[1 - k]
        fwd = (lambda m, d: m(d)) if kind == "node_removal" else (lambda m, d: _six(m, d))   # noqa: E731
        loss, grads, (f_net, qo, action, reward, nonfinal, mine) = _fused_step(net, other, mb, select, gamma, dev, n, 512)
        # autograd reference of the same step
        net.zero_grad(set_to_none=True)
        states = [s for s, _, _, _ in mb]
        nexts = [(nx if nx is not None else s) for s, _, nx, _ in mb]
        bs, bn = Batch.from_data_list(states).to(dev), Batch.from_data_list(nexts).to(dev)
        if select:
            with torch.no_grad():
                nv = fwd(other, bn).max(1)[0] * nonfinal
            pred = fwd(net, bs).gather(1, action.reshape(-1, 1)).squeeze(1)
            ref = torch.nn.HuberLoss()(pred, nv * gamma + reward)
        else:
            with torch.no_grad():
                pred = fwd(other, bs).gather(1, action.reshape(-1, 1)).squeeze(1)
            ref = torch.nn.HuberLoss()(pred, fwd(net, bn).max(1)[0] * nonfinal * gamma + reward)
        ref.backward()
        assert abs(loss - float(ref.detach())) < 2e-5 * max(abs(float(ref.detach())), 1e-3)
        scale = max(float(p.grad.abs().max()) for p in net.parameters() if p.grad is not None)
        assert scale > 1e-7
        for name, p in net.named_parameters():
            if p.grad is None:
                assert float(grads[name].abs().max()) == 0.0, name
            else:
                # relative to the parameter's OWN gradient scale (the deep levels' are orders below the head's)
                own = float(p.grad.abs().max())
                assert float((grads[name] - p.grad).abs().max()) < 5e-4 * max(own, 1e-6 * scale), (kind, select, name)
        # second launch on the same inputs: bitwise the same gradient and loss
        loss2, flat2 = f_net.train_step(*_arrays(mine, dev), n, 512, 0 if select else 1, qo, action, reward, nonfinal, gamma)
        torch.cuda.synchronize()
        off = 0
        for name, p in net.named_parameters():
            assert torch.equal(flat2[off:off + p.numel()].view_as(p), grads[name]), name
            off += p.numel()
        assert float(loss2.item()) == loss


def _six(model, data):
    """AirfoilGCNN.forward on features that are already the two it uses (the module itself slices x[:, [2, 3]])."""
    import torch.nn.functional as F
    from meshdqn_amd.airfoilgcnn import gap, gmp
    x, edge_index, batch = data.x.float(), data.edge_index, data.batch
    outs = []
    for conv, pool in ((model.conv1, model.pool1), (model.conv2, model.pool2), (model.conv3, model.pool3),
                       (model.conv4, model.pool4), (model.conv5, model.pool5), (model.conv6, model.pool6)):
        x = F.relu(conv(x, edge_index))
        x, edge_index, _, batch, _, _ = pool(x, edge_index, None, batch)
        outs.append(torch.cat([gmp(x, batch), gap(x, batch)], dim=1))
    x = outs[0] + outs[1] + outs[2] + outs[3] + outs[4] + outs[5]
    x = F.relu(model.lin1(x))
    x = F.relu(model.lin2(x))
    return model.lin3(x)


def test_pack_kernel_equals_transposed_parameters(lib_built):
    """mdq_gcn_pack: the one-launch repack equals `weight.t().contiguous()` for every segment, and follows an in-place
    parameter update (the optimiser step) without rebuilding the table."""
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from meshdqn_amd.gcn_fused import FusedGcn
    net = NodeRemovalNet(181, conv_width=128, topk=0.1)
    net.set_num_nodes(17)
    net.cuda()
    fg = FusedGcn(net)
    for rep in range(2):
        fg._pack()
        torch.cuda.synchronize()
        segs = [net.conv1.lin_l.weight.t(), net.conv1.lin_l.bias, net.conv1.lin_r.weight.t(), net.pool1.weight.reshape(-1)]
        for buf, want in zip(fg._keep[:4], segs):
            assert torch.equal(buf, want.contiguous().reshape(-1))
        assert torch.equal(fg._keep[-2], net.lin3.weight.t().contiguous().reshape(-1))
        table = fg._table_key
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.25)
        assert rep == 1 or fg._version[:-2] != tuple(p._version for p in net.parameters())
    assert fg._table_key == table


def test_fused_learning_step_edge_cases(lib_built):
    """Ragged minibatch (graphs with fewer nodes than NMAX, a graph without edges, one with a single edge) against
    autograd; and a graph beyond EMAX: NaN loss, and its gradient slice must not leak the previous minibatch's values."""
    from meshdqn_amd.airfoilgcnn import NodeRemovalNet
    from meshdqn_amd.data import Batch, Data
    from meshdqn_amd.gcn_fused import FusedGcn
    dev = torch.device("cuda")
    rng = np.random.default_rng(11)
    torch.manual_seed(7)
    nets = [NodeRemovalNet(181, conv_width=128, topk=0.1) for _ in range(2)]
    for m in nets:
        m.set_num_nodes(17)
        m.to(dev)

    def graph(n, e):
        ei = rng.integers(0, n, size=(2, e)) if e else np.zeros((2, 0), np.int64)
        return Data(x=torch.from_numpy(rng.standard_normal((n, 17))).float(), edge_index=torch.from_numpy(ei).long())
    states = [graph(180, 300), graph(97, 150), graph(180, 0), graph(40, 1), graph(11, 30)]
    nexts = [graph(180, 250), graph(180, 0), graph(60, 90), graph(180, 400), graph(25, 60)]
    action = torch.tensor([3, 180, 0, 17, 99], dtype=torch.int64, device=dev)
    reward = torch.tensor([0.3, -1.0, 0.1, 0.0, 0.7], dtype=torch.float32, device=dev)
    nonfinal = torch.tensor([1.0, 0.0, 1.0, 1.0, 1.0], dtype=torch.float32, device=dev)
    net, other = nets
    f_net, f_other = FusedGcn(net), FusedGcn(other)
    qo = f_other.forward_arrays(*_arrays(nexts, dev), 180, 512)
    loss, flat = f_net.train_step(*_arrays(states, dev), 180, 512, 0, qo, action, reward, nonfinal, 0.9)
    torch.cuda.synchronize()
    flat = flat.clone()
    with torch.no_grad():
        nv = other(Batch.from_data_list(nexts).to(dev)).max(1)[0] * nonfinal
    net.zero_grad(set_to_none=True)
    pred = net(Batch.from_data_list(states).to(dev)).gather(1, action.reshape(-1, 1)).squeeze(1)
    ref = torch.nn.HuberLoss()(pred, nv * 0.9 + reward)
    ref.backward()
    assert abs(float(loss.item()) - float(ref.detach())) < 2e-5 * max(abs(float(ref.detach())), 1e-3)
    scale = max(float(p.grad.abs().max()) for p in net.parameters() if p.grad is not None)
    off = 0
    for name, p in net.named_parameters():
        g = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
        if p.grad is not None:
            own = float(p.grad.abs().max())
            assert float((g - p.grad).abs().max()) < 5e-4 * max(own, 1e-6 * scale), name
    # the same minibatch with graph 0 beyond the edge capacity: NaN loss; the other graphs' gradient only
    loss2, flat2 = f_net.train_step(*_arrays(states, dev), 180, 256, 0, qo, action, reward, nonfinal, 0.9)
    torch.cuda.synchronize()
    assert torch.isnan(loss2).all() and torch.isfinite(flat2).all()
    sub = [1, 2, 3, 4]
    loss3, flat3 = f_net.train_step(*_arrays([states[i] for i in sub], dev), 180, 256, 0, qo[sub].contiguous(), action[sub].contiguous(),
                                    reward[sub].contiguous(), nonfinal[sub].contiguous(), 0.9)
    torch.cuda.synchronize()
    # (mean over 5 graphs vs mean over 4: the refused graph contributes nothing)
    assert torch.allclose(flat2 * 5.0, flat3 * 4.0, rtol=1e-4, atol=1e-5 * float(flat3.abs().max()))
