"""GPU: batched DQN rollout + learning loop on VecEnv2DAirfoil (configs[3] of BASELINE.json, one rank)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_batched_train_loop_runs_and_learns_something(lib_built):
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_vec
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"),
                                geometry_params=dict(mesh=os.path.join(GOLDEN, "ys930.npz")),
                                solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
               agent_params=dict(solver_steps=20, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1,
                                 gt_time=-1, u=-1, p=-1, time_reward=0.005, save_steps=4, goal_vertices=0.95, plot_dir=""))
    ctx = DistContext()
    assert ctx.device.type == "cuda"
    trainer = DQNTrainer(n_actions=180, num_inputs=17, ctx=ctx, batch_size=8, lr=1e-3)
    venv = VecEnv2DAirfoil(cfg, 6, base_env=Env2DAirfoil(cfg), nthreads=2)
    w0 = [p.detach().clone() for p in trainer.policy_net_1.parameters()]
    out = train_loop_vec(trainer, venv, num_steps=5, eps_decay=2)   # fast epsilon decay: greedy actions are exercised
    assert out["rewards"].shape == (5, 6) and out["dones"].shape == (5, 6)
    assert trainer.device_memory is not None and trainer.device_memory.size() == 30 and trainer.memory.size() == 0
    # the optimiser step overlaps the environment step and samples the replay as of the previous step: 0, 6 < 8
    assert len(out["losses"]) == 3 and np.isfinite(out["losses"]).all()
    changed = [not torch.equal(a, b) for a, b in zip(w0, trainer.policy_net_1.parameters()) if b.grad is not None]
    changed2 = any(p.grad is not None for p in trainer.policy_net_2.parameters())
    assert any(changed) or changed2
    # transitions hold per-environment graphs with local node ids
    tr = trainer.device_memory.sample(4).to_transitions()
    for t in tr:
        assert t.state.x.shape == (180, 17) and int(t.state.edge_index.max()) < 180
    # the per-transition replay of lazy references (used when ranks exchange transitions) still works
    trainer2 = DQNTrainer(n_actions=180, num_inputs=17, ctx=ctx, batch_size=8, lr=1e-3)
    out2 = train_loop_vec(trainer2, venv, num_steps=3, eps_decay=2, device_replay=False)
    assert trainer2.device_memory is None and trainer2.memory.size() == 18 and len(out2["losses"]) == 2
    # device replay without the overlap: push first, then optimise
    trainer3 = DQNTrainer(n_actions=180, num_inputs=17, ctx=ctx, batch_size=8, lr=1e-3)
    out3 = train_loop_vec(trainer3, venv, num_steps=3, eps_decay=2, overlap_optimise=False)
    assert trainer3.device_memory.size() == 18 and len(out3["losses"]) == 2


def test_graphed_optimiser_step_equals_eager_step(lib_built):
    """The HIP-graph replay of forward_dense + loss + backward gives the eager path's gradient (same minibatch)."""
    from meshdqn_amd.data import Data
    from meshdqn_amd.trainer import DistContext, DQNTrainer, Transition
    rng = np.random.default_rng(4)

    def graph(e):
        return Data(x=torch.from_numpy(rng.standard_normal((180, 17))).float(),
                    edge_index=torch.from_numpy(rng.integers(0, 180, size=(2, e))).long())

    trs = [Transition(graph(int(rng.integers(200, 600))), torch.tensor([[int(rng.integers(0, 181))]]),
                      None if i % 5 == 0 else graph(int(rng.integers(200, 600))), torch.tensor([float(rng.standard_normal())]))
           for i in range(8)]
    for sel in (True, False):
        grads = []
        for use_graph in (False, True):
            tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(), batch_size=8, lr=0.0)
            tr.graphs = use_graph
            for _ in range(2 if use_graph else 1):   # the second call is a pure replay
                tr.num_grads = 1                      # (not a multiple of target_update: `select` stays as set)
                tr.select = sel
                loss = tr.optimize(trs)
            assert use_graph is False or (0 if sel else 1) in tr._graphs, tr._graph_error
            net = tr.policy_net_1 if sel else tr.policy_net_2
            grads.append((loss, [None if p.grad is None else p.grad.clone() for p in net.parameters()]))
        assert abs(grads[1][0] - grads[0][0]) < 1e-6
        nonzero = 0
        for a, b in zip(grads[1][1], grads[0][1]):
            assert (a is None) == (b is None)      # parameters outside the forward pass keep grad None on both paths
            if a is None:
                continue
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-7)
            nonzero += int(b.abs().max() > 0)
        assert nonzero > 10


def test_lazy_minibatch_path_equals_data_path(lib_built):
    """Optimiser step on a replay of lazy state references (what train_loop_vec stores) vs the same transitions as
    materialised Data objects: same loss, same gradient, for both halves of the double-DQN cycle."""
    from meshdqn_amd.trainer import DistContext, DQNTrainer, StateRef, Transition
    rng = np.random.default_rng(8)
    B, N, F = 8, 180, 17
    counts = rng.integers(150, 500, size=B)
    ep = np.zeros(B + 1, np.int64); ep[1:] = np.cumsum(counts)
    def state():
        return dict(x=torch.from_numpy(rng.standard_normal((B, N, F))).float().cuda(),
                    esrc=torch.from_numpy(rng.integers(0, N, size=int(ep[-1]))).int().cuda(),
                    edst=torch.from_numpy(rng.integers(0, N, size=int(ep[-1]))).int().cuda())
    st0, st1 = state(), state()
    refs0 = [StateRef(st0, b, int(ep[b]), int(ep[b + 1])) for b in range(B)]
    refs1 = [StateRef(st1, b, int(ep[b]), int(ep[b + 1])) for b in range(B)]
    acts = [torch.tensor([[int(rng.integers(0, 181))]]) for _ in range(B)]
    rews = [torch.tensor([float(rng.standard_normal())]) for _ in range(B)]
    lazy = [Transition(refs0[b], acts[b], None if b % 4 == 0 else refs1[b], rews[b]) for b in range(B)]
    data = [Transition(refs0[b].data(), acts[b], None if b % 4 == 0 else refs1[b].data(), rews[b]) for b in range(B)]
    for sel in (True, False):
        res = []
        for trs in (data, lazy):
            tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(), batch_size=B, lr=0.0)
            tr.num_grads, tr.select = 1, sel
            loss = tr.optimize(trs)
            net = tr.policy_net_1 if sel else tr.policy_net_2
            res.append((loss, [None if p.grad is None else p.grad.clone() for p in net.parameters()]))
        assert abs(res[0][0] - res[1][0]) < 1e-6
        for a, b_ in zip(res[0][1], res[1][1]):
            assert (a is None) == (b_ is None)
            assert a is None or torch.allclose(a, b_, rtol=1e-3, atol=1e-7)


def test_device_replay_minibatch_equals_lazy_reference_path(lib_built):
    """A minibatch gathered from the GPU-resident replay ring gives the loss and gradient of the same transitions
    held as lazy state references, for both halves of the double-DQN cycle; ring wrap-around keeps live states."""
    from meshdqn_amd.trainer import DeviceBatch, DeviceReplay, DistContext, DQNTrainer, StateRef, Transition
    rng = np.random.default_rng(21)
    B, N, F, EM = 8, 180, 17, 1536

    def state():
        cnt = rng.integers(150, 500, size=B)
        ep = np.zeros(B + 1, np.int64); ep[1:] = np.cumsum(cnt)
        sp = rng.integers(0, N, size=(B, EM)).astype(np.int32)        # (entries past the count are garbage on purpose)
        dp = rng.integers(0, N, size=(B, EM)).astype(np.int32)
        live = np.arange(EM)[None, :] < cnt[:, None]
        return dict(x=torch.from_numpy(rng.standard_normal((B, N, F))).float().cuda(),
                    edge_src_pad=torch.from_numpy(sp).cuda(), edge_dst_pad=torch.from_numpy(dp).cuda(), nedges=cnt.astype(np.int32),
                    esrc=torch.from_numpy(sp[live]).cuda(), edst=torch.from_numpy(dp[live]).cuda(), ep=ep)

    rep = DeviceReplay(capacity=3 * B, B=B, N=N, F=F, e_max=EM, device=torch.device("cuda"))
    assert rep.K == 5
    states = [state() for _ in range(8)]
    bases = [rep.store(states[0])]
    acts, rews, dones = [], [], []
    for t in range(7):                      # 7 batched steps through a ring of 3 steps of transitions / 5 of states
        bases.append(rep.store(states[t + 1]))
        acts.append(rng.integers(0, 181, size=B)); rews.append(rng.standard_normal(B).astype(np.float32))
        dones.append(rng.random(B) < 0.25)
        rep.push(bases[t], bases[t + 1], acts[-1], rews[-1], dones[-1])
    assert rep.size() == 3 * B
    # the live transitions are those of steps 4, 5, 6: rebuild them as lazy references
    lazy = {}
    for t in (4, 5, 6):
        s0, s1 = states[t], states[t + 1]
        for b in range(B):
            r0 = StateRef(s0, b, int(s0["ep"][b]), int(s0["ep"][b + 1]))
            r1 = None if dones[t][b] else StateRef(s1, b, int(s1["ep"][b]), int(s1["ep"][b + 1]))
            lazy[(bases[t] + b)] = Transition(r0, torch.tensor([[int(acts[t][b])]]), r1, torch.tensor([float(rews[t][b])]))
    idx = rng.permutation(3 * B)[:8]
    devb = DeviceBatch(rep, rep.t_s[idx], rep.t_n[idx], rep.t_a[idx], rep.t_r[idx])
    ref = [lazy[int(sl)] for sl in rep.t_s[idx]]
    for sel in (True, False):
        res = []
        for trs in (ref, devb):
            tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(), batch_size=8, lr=0.0)
            tr.num_grads, tr.select = 1, sel
            loss = tr.optimize(trs)
            assert (0 if sel else 1) in tr._graphs, tr._graph_error
            net = tr.policy_net_1 if sel else tr.policy_net_2
            res.append((loss, [None if p.grad is None else p.grad.clone() for p in net.parameters()]))
        assert abs(res[0][0] - res[1][0]) < 1e-6
        for a, b_ in zip(res[0][1], res[1][1]):
            assert (a is None) == (b_ is None)
            assert a is None or torch.allclose(a, b_, rtol=1e-3, atol=1e-7)
    # eager fallback objects
    for t_, r_ in zip(devb.to_transitions(), ref):
        assert torch.equal(t_.state.x.cpu(), r_.state.x.cpu()) and torch.equal(t_.state.edge_index.cpu(), r_.state.edge_index.cpu())
        assert (t_.next_state is None) == (r_.next_state is None)


@pytest.mark.parametrize("select", [True, False])
def test_graphed_optimiser_step_matches_the_oracle_learning_step(lib_built, select):
    """The HIP-graph replay of the learning step (fused no-grad network + dense autograd path) against the plain-loop
    restatement of the reference's `compute_gradients` (oracle/dqn.py, airfoil_dqn.py:240-310): loss and the gradient
    the optimiser is about to apply."""
    from meshdqn_amd.data import Data
    from meshdqn_amd.trainer import DistContext, DQNTrainer, Transition
    from oracle import gcn as ora
    from oracle.dqn import compute_gradients
    rng = np.random.default_rng(9 + select)
    tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(device=torch.device("cuda")), gamma=0.9, batch_size=8,
                    lr=0.0, weight_decay=0.0)
    oras = []
    for net in (tr.policy_net_1, tr.policy_net_2):
        sd = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)) * 0.06).float() for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        o = ora.NodeRemovalNet(181, conv_width=128, topk=0.1)
        o.set_num_nodes(17)
        o.load_state_dict(sd)
        oras.append(o)

    def graph():
        e = int(rng.integers(200, 500))
        return Data(x=torch.from_numpy(rng.standard_normal((180, 17))).float(),
                    edge_index=torch.from_numpy(rng.integers(0, 180, size=(2, e))).long())
    mb = [(graph(), int(rng.integers(0, 181)), None if i % 3 == 2 else graph(), float(rng.uniform(-1, 1))) for i in range(8)]
    loss_o, grads_o = compute_gradients(oras[0], oras[1], mb, select, tr.gamma)
    # optimize() toggles `select` when num_grads is a multiple of target_update: arrange for the wanted half
    tr.select = not select
    tr.num_grads = 0
    loss = tr.optimize([Transition(s, torch.tensor([[a]]), n, torch.tensor([r])) for s, a, n, r in mb])
    assert tr.select == select and tr._graphs, tr._graph_error
    assert abs(loss - loss_o) < 1e-4 * max(abs(loss_o), 1e-3)
    net = tr.policy_net_1 if select else tr.policy_net_2
    scale = max(float(g.abs().max()) for g in grads_o.values() if g is not None)
    for k, p in net.named_parameters():
        g = grads_o[k]
        if g is None:
            assert p.grad is None, k
        else:
            assert float((p.grad.cpu() - g).abs().max()) < 1e-4 * scale, k


def test_shared_device_replay_minibatch_equals_its_transitions(lib_built):
    """A minibatch sampled from the record ring of the shared replay (what the ranks all-gather) through the HIP-graph
    optimiser step vs the same transitions as `Data` objects on the eager path: same loss, same gradient."""
    import random
    from meshdqn_amd.trainer import DistContext, DQNTrainer, SharedDeviceReplay, pack_transitions_device
    rng = np.random.default_rng(31)
    B, N, F, EM = 8, 180, 17, 1536
    dev = torch.device("cuda")

    def state():
        return dict(x=torch.from_numpy(rng.standard_normal((B, N, F))).float().to(dev),
                    edge_src_pad=torch.from_numpy(rng.integers(0, N, size=(B, EM)).astype(np.int32)).to(dev),
                    edge_dst_pad=torch.from_numpy(rng.integers(0, N, size=(B, EM)).astype(np.int32)).to(dev),
                    nedges=rng.integers(150, 500, size=B).astype(np.int32))
    rep = SharedDeviceReplay(64, N, F, EM, dev)
    for _ in range(3):
        rep.push_records(pack_transitions_device(state(), state(), rng.integers(0, 181, size=B), rng.standard_normal(B),
                                                 rng.random(B) < 0.3, EM))
    for sel in (True, False):
        res = []
        random.seed(4)
        mb = rep.sample(8)
        for trs in (mb, mb.to_transitions()):
            tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(), batch_size=8, lr=0.0)
            tr.graphs = trs is mb
            tr.num_grads, tr.select = 1, sel
            loss = tr.optimize(trs)
            assert trs is not mb or (0 if sel else 1) in tr._graphs, tr._graph_error
            net = tr.policy_net_1 if sel else tr.policy_net_2
            res.append((loss, [None if p.grad is None else p.grad.clone() for p in net.parameters()]))
        assert abs(res[0][0] - res[1][0]) < 1e-6
        for a, b_ in zip(res[0][1], res[1][1]):
            assert (a is None) == (b_ is None)
            assert a is None or torch.allclose(a, b_, rtol=1e-3, atol=1e-7)


def _rccl_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import numpy as np
    import torch
    from meshdqn_amd.data import Data
    from meshdqn_amd.trainer import (DistContext, DQNTrainer, Transition, allgather_records, pack_transitions_device)
    ctx = DistContext(backend="nccl", device=torch.device("cuda", rank))
    tr = DQNTrainer(180, 17, ctx=ctx, lr=1e-3, target_update=2, batch_size=8)
    rng = np.random.default_rng(50 + rank)

    def graph():
        e = int(rng.integers(200, 500))
        return Data(x=torch.from_numpy(rng.standard_normal((180, 17))).float(),
                    edge_index=torch.from_numpy(rng.integers(0, 180, size=(2, e))).long())
    losses = []
    for step in range(3):
        trs = [Transition(graph(), torch.tensor([[int(rng.integers(0, 181))]]), None if i % 4 == 3 else graph(),
                          torch.tensor([float(rng.standard_normal())])) for i in range(8)]
        losses.append(tr.optimize(trs))
    w1 = torch.cat([p.detach().reshape(-1) for p in tr.policy_net_1.parameters()]).cpu().numpy()
    w2 = torch.cat([p.detach().reshape(-1) for p in tr.policy_net_2.parameters()]).cpu().numpy()
    rec = torch.full((4, 7), float(rank), device=ctx.device)
    allrec = allgather_records(ctx, rec).cpu().numpy()
    q.put((rank, w1, w2, losses, allrec))
    ctx.barrier()
    ctx.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_gpu_rccl_gradient_allreduce_and_record_allgather(lib_built):
    """The first multi-GPU lease exercises RCCL itself: two ranks on two GPUs take three optimiser steps on different
    data (HIP-graph path, one flat all-reduce each) and must end with identical replicas; one all-gather of records."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert res[0][3] != res[1][3]
    assert np.array_equal(res[0][4], res[1][4]) and res[0][4][:4].max() == 0.0 and res[0][4][4:].min() == 1.0


def _rccl_single_worker(port, q):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      MDQ_FORCE_COLLECTIVES="1")
    import numpy as np
    import torch
    import torch.distributed as dist
    from meshdqn_amd.data import Data
    from meshdqn_amd.trainer import (DistContext, DQNTrainer, Transition, allgather_records, allgather_records_into)
    ctx = DistContext(backend="nccl", device=torch.device("cuda", 0))
    info = dict(multi=ctx.multi, backend=ctx.backend, initialized=dist.is_initialized(), world=dist.get_world_size())
    tr = DQNTrainer(180, 17, ctx=ctx, lr=1e-3, target_update=2, batch_size=8)
    rng = np.random.default_rng(50)

    def graph():
        e = int(rng.integers(200, 500))
        return Data(x=torch.from_numpy(rng.standard_normal((180, 17))).float(),
                    edge_index=torch.from_numpy(rng.integers(0, 180, size=(2, e))).long())
    losses = []
    for step in range(3):
        trs = [Transition(graph(), torch.tensor([[int(rng.integers(0, 181))]]), None if i % 4 == 3 else graph(),
                          torch.tensor([float(rng.standard_normal())])) for i in range(8)]
        losses.append(tr.optimize(trs))
    info["graph_path"] = sorted(tr._graphs) if hasattr(tr, "_graphs") else None
    rec = torch.arange(28, dtype=torch.float32, device=ctx.device).reshape(4, 7)
    allrec = allgather_records(ctx, rec).cpu().numpy()
    ring = torch.zeros((8, 7), dtype=torch.float32, device=ctx.device)
    ring[4:8] = rec
    allgather_records_into(ctx, ring, 4, 4, 4)              # the in-place form of the device loop (group of W = world * B rows)
    torch.cuda.synchronize()
    info["max_over_ranks"] = ctx.max_over_ranks(1.25)
    q.put((info, losses, allrec, ring.cpu().numpy()))
    ctx.barrier()
    ctx.close()


def test_single_rank_rccl_process_group(lib_built):
    """What a one-GPU box can say about the RCCL path: `MDQ_FORCE_COLLECTIVES=1` makes a job of ONE rank create its process
    group (backend "nccl" = RCCL, communicator on the device) and run every collective of the multi-GPU job through it -
    the flat gradient all-reduce inside the optimiser step (three steps, the captured-graph path where it is taken), both
    forms of the record all-gather, the timing reduction and the barrier - with the numbers of a job without a group.
    (Two ranks need two GPUs: the test above.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    p = ctxm.Process(target=_rccl_single_worker, args=(port, q))
    p.start()
    info, losses, allrec, ring = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert info["multi"] and info["backend"] == "nccl" and info["initialized"] and info["world"] == 1, info
    assert info["max_over_ranks"] == 1.25
    want = np.arange(28, dtype=np.float32).reshape(4, 7)
    assert np.array_equal(allrec, want) and np.array_equal(ring[4:8], want) and (ring[:4] == 0).all()
    # the same three steps without a process group
    from meshdqn_amd.data import Data
    from meshdqn_amd.trainer import DistContext, DQNTrainer, Transition
    tr = DQNTrainer(180, 17, ctx=DistContext(), lr=1e-3, target_update=2, batch_size=8)
    assert not tr.ctx.multi
    rng = np.random.default_rng(50)

    def graph():
        e = int(rng.integers(200, 500))
        return Data(x=torch.from_numpy(rng.standard_normal((180, 17))).float(),
                    edge_index=torch.from_numpy(rng.integers(0, 180, size=(2, e))).long())
    ref = []
    for step in range(3):
        trs = [Transition(graph(), torch.tensor([[int(rng.integers(0, 181))]]), None if i % 4 == 3 else graph(),
                          torch.tensor([float(rng.standard_normal())])) for i in range(8)]
        ref.append(tr.optimize(trs))
    assert np.allclose(ref, losses, rtol=1e-5, atol=1e-7), (ref, losses, info)
