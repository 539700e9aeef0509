"""CPU tests of the data-parallel trainer: world_size-2 gloo gradient all-reduce, replay all-gather,
environment sharding, epsilon schedule, double-DQN bookkeeping."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from meshdqn_amd.data import Data
from meshdqn_amd.trainer import (DQNTrainer, DistContext, ReplayMemory, Transition, allgather_transitions,
                                 epsilon_threshold, pack_transitions, unpack_transitions)


def _state(rng, n=180, f=17, e=372):
    return Data(x=torch.from_numpy(rng.standard_normal((n, f))).float(),
                edge_index=torch.from_numpy(rng.integers(0, n, size=(2, e))).long(), edge_attr=[])


def _transitions(seed, count=8):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        done = (i % 4 == 3)
        out.append(Transition(_state(rng), torch.tensor([[int(rng.integers(0, 181))]]), None if done else _state(rng),
                              torch.tensor([float(rng.standard_normal())])))
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    ctx = DistContext(backend="gloo", device=torch.device("cpu"))
    tr = DQNTrainer(180, 17, ctx=ctx, lr=1e-3, target_update=2)
    p0 = tr.policy_net_2.flat_gradients().numel()
    w0 = torch.cat([p.detach().reshape(-1) for p in tr.policy_net_1.parameters()]).clone()
    losses = []
    for step in range(3):
        losses.append(tr.optimize(_transitions(100 * step + rank)))
    w1 = torch.cat([p.detach().reshape(-1) for p in tr.policy_net_1.parameters()])
    w2 = torch.cat([p.detach().reshape(-1) for p in tr.policy_net_2.parameters()])
    # replay all-gather of one transition per rank
    got = allgather_transitions(ctx, _transitions(7 + rank, 1), 180, 17, 512)
    shard = list(ctx.shard(11))
    tmax = ctx.max_over_ranks(1.0 + rank)
    q.put((rank, p0, w0.numpy(), w1.numpy(), w2.numpy(), losses, len(got), float(got[1].state.x.sum()), shard, tmax,
           tr.select, tr.num_grads))
    ctx.barrier()
    ctx.close()


def test_two_rank_gloo_allreduce_and_gather():
    world, port = 2, _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0[1] == 173493
    # replicas start identical, stay identical after 3 all-reduced steps, and did move
    assert np.array_equal(r0[2], r1[2])
    assert np.array_equal(r0[3], r1[3]) and np.array_equal(r0[4], r1[4])
    assert not np.array_equal(r0[2], r0[3])
    # the ranks saw different data (different local losses) - only the all-reduce keeps them in sync
    assert r0[5] != r1[5]
    assert r0[6] == r1[6] == 2
    exp = float(_transitions(8, 1)[0].state.x.sum())
    assert abs(r0[7] - exp) < 1e-3 and abs(r1[7] - exp) < 1e-3
    assert r0[8] == [0, 1, 2, 3, 4, 5] and r1[8] == [6, 7, 8, 9, 10]
    assert r0[9] == r1[9] == 2.0
    assert r0[10] == r1[10] and r0[11] == r1[11] == 3


def test_allreduced_gradient_is_the_rank_mean():
    """Single process check of what the 2-rank run must produce: averaging the two local gradients."""
    torch.manual_seed(0)
    tr = DQNTrainer(180, 17, ctx=DistContext(device=torch.device("cpu")), lr=1e-3)
    grads = []
    for rank in range(2):
        tr.select = True
        tr.policy_net_1.zero_grad(set_to_none=True)
        tr._loss(_transitions(rank)).backward()
        grads.append(tr.policy_net_1.flat_gradients().clone())
    mean = 0.5 * (grads[0] + grads[1])
    tr.policy_net_1.set_flat_gradients(mean)
    assert torch.allclose(tr.policy_net_1.flat_gradients(), mean)
    assert mean.abs().sum() > 0


def test_transition_pack_roundtrip_and_replay_ring():
    trs = _transitions(3, 6)
    rec = pack_transitions(trs, 180, 17, 512)
    assert rec.shape == (6, 2 * 180 * 17 + 4 * 512 + 5)
    back = unpack_transitions(rec, 180, 17, 512)
    for a, b in zip(trs, back):
        assert torch.equal(a.state.x, b.state.x) and torch.equal(a.state.edge_index, b.state.edge_index)
        assert (a.next_state is None) == (b.next_state is None)
        if a.next_state is not None:
            assert torch.equal(a.next_state.edge_index, b.next_state.edge_index)
        assert int(a.action) == int(b.action) and abs(float(a.reward) - float(b.reward)) < 1e-6
    mem = ReplayMemory(4)
    for i in range(6):
        mem.push(i, i, i, i)
    assert mem.size() == 4 and sorted(t.state for t in mem.memory) == [2, 3, 4, 5]
    assert len(mem.sample(3)) == 3
    assert abs(epsilon_threshold(0) - 1.0) < 1e-12 and abs(epsilon_threshold(10000) - (0.01 + 0.99 / np.e)) < 1e-12


def test_double_dqn_select_toggles_every_target_update():
    tr = DQNTrainer(180, 17, ctx=DistContext(device=torch.device("cpu")), target_update=2)
    seen = []
    for i in range(5):
        tr.optimize(_transitions(i, 4))
        seen.append(tr.select)
    # toggled when num_grads % 2 == 0 BEFORE the step (airfoil_dqn.py:185-186), starting from True
    assert seen == [False, False, True, True, False]


def test_training_log_files_and_restart(tmp_path):
    """reward / rewards / losses / actions / eps .npy with the reference's semantics (airfoil_dqn.py:79-133)."""
    from meshdqn_amd.trainer import TrainingLog
    d = str(tmp_path)
    log = TrainingLog(d)
    log.add_eps(0.9)
    log.add_loss(0.25)
    log.add_episode([1.0, -1.0], [3, 180])
    log.add_episode([0.5], [7])
    log.write()
    assert np.load(os.path.join(d, "reward.npy")).tolist() == [0.0, 0.5]
    again = TrainingLog(d, restart=True)
    assert again.rewards == [0.0, 0.5] and [list(a) for a in again.actions] == [[3, 180], [7]]
    again.add_episode([2.0], [1])
    again.write()
    assert np.load(os.path.join(d, "RESTART_reward.npy")).tolist() == [0.0, 0.5, 2.0]


def test_device_replay_ring_on_cpu_tensors():
    """DeviceReplay bookkeeping (pure torch ops, here on CPU tensors): slots of states / next states across the ring
    wrap-around, terminal flags, compacted and padded edge arrays of a gathered minibatch."""
    import numpy as np
    from meshdqn_amd.trainer import DeviceReplay
    rng = np.random.default_rng(3)
    B, N, F, EM = 4, 6, 3, 16

    def state(tag):
        cnt = rng.integers(1, EM + 1, size=B)
        sp = rng.integers(0, N, size=(B, EM)).astype(np.int32)
        dp = rng.integers(0, N, size=(B, EM)).astype(np.int32)
        return dict(x=torch.full((B, N, F), float(tag)) + torch.arange(B).reshape(B, 1, 1), edge_src_pad=torch.from_numpy(sp),
                    edge_dst_pad=torch.from_numpy(dp), nedges=cnt.astype(np.int32))

    rep = DeviceReplay(capacity=2 * B, B=B, N=N, F=F, e_max=EM, device=torch.device("cpu"))
    assert rep.K == 4 and DeviceReplay.eligible(state(0), EM) and not DeviceReplay.eligible(dict(x=None), EM)
    states = [state(t) for t in range(7)]
    bases = [rep.store(states[0])]
    for t in range(6):
        bases.append(rep.store(states[t + 1]))
        rep.push(bases[t], bases[t + 1], np.full(B, t), np.full(B, 0.5 * t, np.float32), np.arange(B) == t % B)
    assert rep.size() == 2 * B and bases == [0, 4, 8, 12, 0, 4, 8]
    # the live transitions are those of steps 4 and 5; their states are still in the ring
    live = {}
    for t in (4, 5):
        for b in range(B):
            live[(t, b)] = (bases[t] + b, -1 if b == t % B else bases[t + 1] + b)
    got = sorted(zip(rep.t_s.tolist(), rep.t_n.tolist(), rep.t_a.tolist()))
    assert got == sorted((s, n, t) for (t, b), (s, n) in live.items())
    slots = np.array([bases[5] + 2, bases[6] + 1, bases[4] + 0])
    g = rep.gather(slots)
    src_states = [(states[5], 2), (states[6], 1), (states[4], 0)]
    off = 0
    for i, (st, b) in enumerate(src_states):
        c = int(st["nedges"][b])
        assert torch.equal(g["x"][i], st["x"][b])
        assert torch.equal(g["esrc"][off:off + c], st["edge_src_pad"][b, :c]) and torch.equal(g["edst"][off:off + c], st["edge_dst_pad"][b, :c])
        assert torch.equal(g["src"][i, :c], st["edge_src_pad"][b, :c].long()) and int(g["src"][i, c:].abs().sum()) == 0
        assert g["mask"][i].sum().item() == c and int(g["edge_ptr"][i]) == off
        off += c
    assert int(g["edge_ptr"][-1]) == off and g["node_ptr"].tolist() == [0, N, 2 * N, 3 * N]
    d = rep.data(int(slots[0]))
    assert d.x.shape == (N, F) and d.edge_index.shape == (2, int(states[5]["nedges"][2]))


def _batched_state(rng, B=4, N=180, F=17, EM=512):
    cnt = rng.integers(50, EM, size=B).astype(np.int32)
    return dict(x=torch.from_numpy(rng.standard_normal((B, N, F))).float(),
                edge_src_pad=torch.from_numpy(rng.integers(0, N, size=(B, EM)).astype(np.int32)),
                edge_dst_pad=torch.from_numpy(rng.integers(0, N, size=(B, EM)).astype(np.int32)), nedges=cnt)


def _state_data(st, b):
    c = int(st["nedges"][b])
    return Data(x=st["x"][b].clone(), edge_index=torch.stack([st["edge_src_pad"][b, :c].long(), st["edge_dst_pad"][b, :c].long()]),
                edge_attr=[])


def test_device_packing_equals_host_packing_and_feeds_the_shared_replay():
    """`pack_transitions_device` (vectorised, no per-field host copies) writes exactly the records of `pack_transitions`;
    `SharedDeviceReplay` gives the same minibatch arrays as the transitions they were packed from."""
    from meshdqn_amd.trainer import SharedDeviceReplay, pack_transitions_device
    rng = np.random.default_rng(5)
    B, N, F, EM = 4, 180, 17, 512
    st0, st1 = _batched_state(rng), _batched_state(rng)
    acts = rng.integers(0, 181, size=B)
    rews = rng.standard_normal(B).astype(np.float32)
    dones = np.array([False, True, False, False])
    rec = pack_transitions_device(st0, st1, acts, rews, dones, EM)
    trs = [Transition(_state_data(st0, b), torch.tensor([[int(acts[b])]]), None if dones[b] else _state_data(st1, b),
                      torch.tensor([float(rews[b])])) for b in range(B)]
    assert torch.equal(rec, pack_transitions(trs, N, F, EM))
    rep = SharedDeviceReplay(16, N, F, EM, torch.device("cpu"))
    rep.push_records(rec)
    rep.push_records(rec)
    assert rep.size() == 8
    import random
    random.seed(0)
    mb = rep.sample(8)
    back = mb.to_transitions()
    for t in back:
        # every sampled transition is one of the four originals
        match = [o for o in trs if torch.equal(o.state.x, t.state.x)]
        assert len(match) == 1
        o = match[0]
        assert torch.equal(o.state.edge_index, t.state.edge_index) and int(o.action) == int(t.action)
        assert (o.next_state is None) == (t.next_state is None)
        if o.next_state is not None:
            assert torch.equal(o.next_state.x, t.next_state.x) and torch.equal(o.next_state.edge_index, t.next_state.edge_index)
    # ring wrap-around keeps the newest records
    for _ in range(3):
        rep.push_records(rec)
    assert rep.size() == 16 and rep.position == (5 * B) % 16


def _gather_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from meshdqn_amd.trainer import SharedDeviceReplay, allgather_records, pack_transitions_device
    ctx = DistContext(backend="gloo", device=torch.device("cpu"))
    rng = np.random.default_rng(100 + rank)
    st0, st1 = _batched_state(rng), _batched_state(rng)
    rec = pack_transitions_device(st0, st1, np.full(4, rank), np.full(4, 0.5 * rank, np.float32), np.zeros(4, bool), 512)
    allrec = allgather_records(ctx, rec)
    rep = SharedDeviceReplay(32, 180, 17, 512, torch.device("cpu"))
    rep.push_records(allrec)
    q.put((rank, tuple(allrec.shape), float(allrec[:, -3].sum()), float(allrec.sum()), rep.size()))
    ctx.barrier()
    ctx.close()


def test_two_rank_record_allgather():
    world, port = 2, _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # both ranks hold the same 8 records (4 of each rank, rank order); the action column sums to 4 * 0 + 4 * 1
    assert res[0][1] == res[1][1] == (8, 2 * 180 * 17 + 4 * 512 + 5)
    assert res[0][2] == res[1][2] == 4.0 and res[0][3] == res[1][3] and res[0][4] == res[1][4] == 8


def test_per_worker_loop_with_several_ranks_needs_a_common_step_count():
    from meshdqn_amd.trainer import train_loop_per_worker
    tr = DQNTrainer(180, 17, ctx=DistContext(device=torch.device("cpu")))
    tr.ctx.world = 2            # (no process group needed: the check comes first)
    with pytest.raises(ValueError, match="max_steps"):
        train_loop_per_worker(tr, lambda: None, num_episodes=3)


def test_trainer_save_load_roundtrip(tmp_path):
    """RESTART (airfoil_dqn.py:163-179): both networks, optimiser / scheduler state, gradient count, toggle and the
    caller's extras come back; reference-style checkpoints (two .pt files only) load too."""
    tr = DQNTrainer(180, 17, ctx=DistContext(device=torch.device("cpu")), lr=1e-3, target_update=2)
    for i in range(3):
        tr.optimize(_transitions(i, 4))
    tr.save(str(tmp_path), "restart_", extra=dict(steps_done=np.arange(5)))
    tr2 = DQNTrainer(180, 17, ctx=DistContext(device=torch.device("cpu")), lr=1e-3, target_update=2, seed=99)
    extra = tr2.load(str(tmp_path), "restart_")
    assert extra["steps_done"].tolist() == [0, 1, 2, 3, 4] and tr2.num_grads == 3 and tr2.select == tr.select
    for a, b in zip(tr.policy_net_1.state_dict().values(), tr2.policy_net_1.state_dict().values()):
        assert torch.equal(a, b)
    # the next optimiser step is identical on both (Adam moments restored)
    l1, l2 = tr.optimize(_transitions(7, 4)), tr2.optimize(_transitions(7, 4))
    assert abs(l1 - l2) < 1e-7
    for a, b in zip(tr.policy_net_2.parameters(), tr2.policy_net_2.parameters()):
        assert torch.allclose(a, b, atol=1e-7)
    os.remove(os.path.join(str(tmp_path), "restart_trainer_state.pt"))
    tr3 = DQNTrainer(180, 17, ctx=DistContext(device=torch.device("cpu")))
    assert tr3.load(str(tmp_path), "restart_", scheduler_steps=10) == {} and tr3.scheds[0].last_epoch == 10


def _forced_worker(port, q):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      MDQ_FORCE_COLLECTIVES="1")
    torch.set_num_threads(1)
    import torch.distributed as dist
    ctx = DistContext(backend="gloo", device=torch.device("cpu"))
    tr = DQNTrainer(180, 17, ctx=ctx, lr=1e-3, target_update=2)
    losses = [tr.optimize(_transitions(100 * step)) for step in range(3)]
    got = allgather_transitions(ctx, _transitions(7, 1), 180, 17, 512)
    w1 = torch.cat([p.detach().reshape(-1) for p in tr.policy_net_1.parameters()])
    q.put((ctx.multi, ctx.backend, dist.is_initialized() and dist.get_world_size(), losses, len(got), w1.numpy(), ctx.max_over_ranks(2.5)))
    ctx.barrier()
    ctx.close()


def test_one_rank_with_forced_collectives_equals_no_process_group():
    """`MDQ_FORCE_COLLECTIVES=1`: ONE rank creates its process group and sends the gradient all-reduce, the transition
    all-gather, the timing reduction and the barrier through the backend (gloo here; "nccl" = RCCL on a GPU box:
    tests/test_trainer_gpu.py) - with the numbers of a trainer that has no group at all."""
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    p = ctxm.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    multi, backend, world, losses, ngot, w1, tmax = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert multi and backend == "gloo" and world == 1 and ngot == 1 and tmax == 2.5
    os.environ.pop("MDQ_FORCE_COLLECTIVES", None)
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)                  # (as the worker: Adam turns the round-off of a threaded reduction into O(lr) steps)
    try:
        tr = DQNTrainer(180, 17, ctx=DistContext(device=torch.device("cpu")), lr=1e-3, target_update=2)
        assert not tr.ctx.multi
        ref = [tr.optimize(_transitions(100 * step)) for step in range(3)]
        w_ref = torch.cat([p_.detach().reshape(-1) for p_ in tr.policy_net_1.parameters()]).numpy()
    finally:
        torch.set_num_threads(nthr)
    assert np.allclose(ref, losses, rtol=1e-6, atol=1e-8)
    assert np.allclose(w_ref, w1, rtol=1e-5, atol=1e-7)
