"""Oracle: the learning step of the reference (`DataWorker._get_data` + `compute_gradients`, airfoil_dqn.py:240-310)
as a plain per-transition loop over the oracle networks (oracle/gcn.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED like oracle/gcn.py (the reference holds no vector
for the learning step).  Restated semantics:

  select True : prediction  q1(s)[a] WITH gradient through policy_net_1; target from policy_net_2 without gradient
  select False: prediction  q1(s)[a] without gradient; the gradient flows through policy_net_2 in the target
                max_a' q2(s')[a'] (the reference's toggle, :256-259 and :270-274)
  terminal transitions (next_state None) have next value 0 (:264,:276)
  expected = next_value * GAMMA + reward (:279);  loss = HuberLoss(pred, expected), delta 1, mean (:299-300)
  returned gradients are those of the SELECTED network (:303-306), here as a dict by parameter name
"""
import torch


def huber_mean(pred, target):
    """torch.nn.HuberLoss() with its defaults (delta = 1, reduction 'mean'), written out."""
    total = 0.0
    for p, t in zip(pred, target):
        d = p - t
        total = total + (0.5 * d * d if abs(float(d.detach())) <= 1.0 else abs(d) - 0.5)
    return total / len(pred)


def compute_gradients(net1, net2, transitions, select, gamma):
    """transitions: list of (state, action:int, next_state or None, reward:float); states are graph objects with
    x / edge_index (batch None).  Returns (loss value, {parameter name: gradient} of the selected network)."""
    net = net1 if select else net2
    for p in net.parameters():
        p.grad = None
    pred, target = [], []
    for state, action, next_state, reward in transitions:
        if select:
            out = net1(state)
        else:
            with torch.no_grad():
                out = net1(state)
        pred.append(out[0, int(action)].float())
        nxt = torch.zeros(())
        if next_state is not None:
            if select:
                with torch.no_grad():
                    nxt = net2(next_state).max(1)[0][0].float()
            else:
                nxt = net2(next_state).max(1)[0][0].float()
        target.append(nxt * gamma + float(reward))
    loss = huber_mean(pred, target)
    if loss.requires_grad:
        loss.backward()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    return float(loss.detach()), grads
