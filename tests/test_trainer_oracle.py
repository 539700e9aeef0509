"""CPU: the optimiser step of `DQNTrainer` (loss + gradient of the double-DQN update, both halves of the `select`
toggle) against the plain-loop restatement of the reference's learning step in oracle/dqn.py (airfoil_dqn.py:240-310)."""
import numpy as np
import pytest
import torch


def _minibatch(rng, n=8, nodes=180, feat=17):
    from meshdqn_amd.data import Data
    out = []
    for i in range(n):
        def graph():
            e = int(rng.integers(200, 500))
            return Data(x=torch.from_numpy(rng.standard_normal((nodes, feat))).float(),
                        edge_index=torch.from_numpy(rng.integers(0, nodes, size=(2, e))).long())
        out.append((graph(), int(rng.integers(0, 181)), None if i % 3 == 2 else graph(), float(rng.uniform(-1, 1))))
    return out


def _nets(rng):
    from meshdqn_amd.trainer import DistContext, DQNTrainer
    from oracle import gcn as ora
    tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(device=torch.device("cpu")), gamma=0.9)
    oras = []
    for net in (tr.policy_net_1, tr.policy_net_2):
        # small weights: the softmax head must not saturate (a one-hot output has a zero gradient in fp32)
        sd = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)) * 0.06).float() for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        o = ora.NodeRemovalNet(181, conv_width=128, topk=0.1)
        o.set_num_nodes(17)
        o.load_state_dict(sd)
        oras.append(o)
    return tr, oras


@pytest.mark.parametrize("select", [True, False])
@pytest.mark.parametrize("dense", [True, False])
def test_loss_and_gradient_match_the_oracle_learning_step(select, dense):
    from meshdqn_amd.trainer import Transition
    from oracle.dqn import compute_gradients
    rng = np.random.default_rng(5 + select)
    tr, (o1, o2) = _nets(rng)
    tr.dense = dense
    mb = _minibatch(rng)
    loss_o, grads_o = compute_gradients(o1, o2, mb, select, tr.gamma)
    tr.select = select
    net = tr.policy_net_1 if select else tr.policy_net_2
    net.zero_grad(set_to_none=True)
    loss = tr._loss([Transition(s, torch.tensor([[a]]), n, torch.tensor([r])) for s, a, n, r in mb])
    loss.backward()
    assert abs(float(loss.detach()) - loss_o) < 1e-6 * max(abs(loss_o), 1e-3)
    scale = max(float(g.abs().max()) for g in grads_o.values() if g is not None)
    assert scale > 0
    used = 0
    for k, p in net.named_parameters():
        g = grads_o[k]
        if g is None:       # conv3 / pool3 / conv6 / pool6 never enter the forward pass (airfoilgcnn.py:106-110,124-128)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        used += 1
        assert p.grad is not None, k
        assert float((p.grad - g).abs().max()) < 2e-5 * scale, (k, float((p.grad - g).abs().max()), scale)
    assert used >= 15
    assert {k for k, g in grads_o.items() if g is None} == {k for k, _ in net.named_parameters()
                                                            if k.split(".")[0] in ("conv3", "pool3", "conv6", "pool6")}


def test_huber_written_out_equals_torch():
    from oracle.dqn import huber_mean
    rng = np.random.default_rng(0)
    p = torch.from_numpy(rng.standard_normal(50) * 2).float()
    t = torch.from_numpy(rng.standard_normal(50) * 2).float()
    assert abs(float(huber_mean(list(p), list(t))) - float(torch.nn.HuberLoss()(p, t))) < 1e-6
