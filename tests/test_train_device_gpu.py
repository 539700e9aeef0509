"""GPU: the device-resident learning loop - replay ring kernels (mdq_replay_step / mdq_replay_sample), the Adam kernel
(mdq_adam_step) and `train_loop_device` against their torch / host-loop counterparts."""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _cfg():
    return dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"),
                                 geometry_params=dict(mesh=os.path.join(GOLDEN, "ys930.npz")),
                                 solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
                agent_params=dict(solver_steps=20, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1,
                                  gt_time=-1, u=-1, p=-1, time_reward=0.005, save_steps=4, goal_vertices=0.95, plot_dir=""))


def test_replay_kernels_match_the_torch_record_packing(lib_built):
    """mdq_replay_step (two halves per batched state) writes the records `pack_transitions_device` builds from the
    two state dicts; mdq_replay_sample returns the minibatch arrays of `SharedDeviceReplay.sample`."""
    from meshdqn_amd import _lib
    from meshdqn_amd.trainer import SharedDeviceReplay, pack_transitions_device
    lib = _lib.load()
    dev = torch.device("cuda")
    rng = np.random.default_rng(3)
    B, N, F, EM = 5, 12, 3, 40

    def state():
        cnt = rng.integers(0, EM, B).astype(np.int32)
        return dict(x=torch.from_numpy(rng.standard_normal((B, N, F))).float().to(dev),
                    edge_src_pad=torch.from_numpy(rng.integers(0, N, (B, EM)).astype(np.int32)).to(dev),
                    edge_dst_pad=torch.from_numpy(rng.integers(0, N, (B, EM)).astype(np.int32)).to(dev),
                    nedges=cnt, nedges_dev=torch.from_numpy(cnt).to(dev))
    states = [state() for _ in range(4)]
    acts = [torch.from_numpy(rng.integers(0, N + 1, B).astype(np.int32)).to(dev) for _ in range(3)]
    rews = [torch.from_numpy(rng.standard_normal(B)).to(dev) for _ in range(3)]
    dones = [torch.from_numpy((rng.random(B) < 0.4).astype(np.uint8)).to(dev) for _ in range(3)]
    rep = SharedDeviceReplay(4 * B, N, F, EM, dev)
    for k, st in enumerate(states):
        _lib.check(lib.mdq_replay_step(rep.R.data_ptr(), rep.rec_len, rep.capacity, B, N * F, EM, st["x"].data_ptr(),
                                       st["edge_src_pad"].data_ptr(), st["edge_dst_pad"].data_ptr(), st["nedges_dev"].data_ptr(),
                                       k * B, (k - 1) * B if k else -1, acts[k - 1].data_ptr() if k else None,
                                       rews[k - 1].data_ptr() if k else None, dones[k - 1].data_ptr() if k else None, None),
                   "mdq_replay_step")
    torch.cuda.synchronize()
    for k in range(3):
        want = pack_transitions_device(states[k], states[k + 1], acts[k], rews[k], dones[k], EM)
        assert torch.equal(rep.R[k * B:(k + 1) * B], want), k
    # sampling: the kernel's arrays against the torch gather of SharedDeviceReplay.sample
    rep.count = 3 * B
    idx = np.array([7, 0, 14, 3, 9, 11], np.int32)
    rows = rep.R.index_select(0, torch.from_numpy(idx.astype(np.int64)).to(dev))
    nf, off = N * F, 2 * N * F + 4 * EM
    done = rows[:, off + 4] > 0.5
    mb = len(idx)

    class SampleDesc(C.Structure):
        _fields_ = [("n", C.c_int32), ("rec_len", C.c_int32), ("nf", C.c_int32), ("EM", C.c_int32)] + \
                   [(nm, C.c_void_p) for nm in ("R", "idx", "x_s", "x_n", "esrc_s", "edst_s", "esrc_n", "edst_n", "edge_ptr_s",
                                                "edge_ptr_n", "action", "reward", "nonfinal")]
    i32 = torch.int32
    out = dict(idx=torch.from_numpy(idx).to(dev), x_s=torch.zeros((mb, N, F), device=dev), x_n=torch.zeros((mb, N, F), device=dev),
               esrc_s=torch.zeros(mb * EM, dtype=i32, device=dev), edst_s=torch.zeros(mb * EM, dtype=i32, device=dev),
               esrc_n=torch.zeros(mb * EM, dtype=i32, device=dev), edst_n=torch.zeros(mb * EM, dtype=i32, device=dev),
               edge_ptr_s=torch.zeros(mb + 1, dtype=i32, device=dev), edge_ptr_n=torch.zeros(mb + 1, dtype=i32, device=dev),
               action=torch.zeros(mb, dtype=torch.int64, device=dev), reward=torch.zeros(mb, device=dev),
               nonfinal=torch.zeros(mb, device=dev))
    d = SampleDesc()
    d.n, d.rec_len, d.nf, d.EM, d.R = mb, rep.rec_len, nf, EM, rep.R.data_ptr()
    for k_, v_ in out.items():
        setattr(d, k_, v_.data_ptr())
    _lib.check(lib.mdq_replay_sample(C.byref(d), None), "mdq_replay_sample")
    torch.cuda.synchronize()
    nxt = rows.clone()
    nxt[:, nf:2 * nf] = torch.where(done[:, None], rows[:, :nf], rows[:, nf:2 * nf])
    nxt[:, 2 * nf + 2 * EM:off] = torch.where(done[:, None], rows[:, 2 * nf:2 * nf + 2 * EM], rows[:, 2 * nf + 2 * EM:off])
    nxt[:, off + 1] = torch.where(done, rows[:, off], rows[:, off + 1])
    ga, gb = rep._graphs(rows, 0), rep._graphs(nxt, 1)
    for g, sfx in ((ga, "_s"), (gb, "_n")):
        ne = int(g["edge_ptr"][-1])
        assert torch.equal(out["x" + sfx], g["x"])
        assert torch.equal(out["edge_ptr" + sfx], g["edge_ptr"])
        assert torch.equal(out["esrc" + sfx][:ne], g["esrc"][:ne]) and torch.equal(out["edst" + sfx][:ne], g["edst"][:ne])
    assert torch.equal(out["action"], rows[:, off + 2].long()) and torch.equal(out["reward"], rows[:, off + 3])
    assert torch.equal(out["nonfinal"], (~done).float())


def test_adam_kernel_equals_torch_adam(lib_built):
    """mdq_adam_step through DQNTrainer._adam_step_device against torch.optim.Adam (weight decay, bias corrections,
    MultiStepLR) over several steps; unused parameters stay untouched and without optimiser state; the torch
    optimiser's `state_dict` carries the moments (checkpoints), and torch's own `step()` continues from them."""
    from meshdqn_amd.trainer import DistContext, DQNTrainer
    dev = torch.device("cuda")
    tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(device=dev), lr=3e-3, weight_decay=1e-2)
    ref = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(device=dev), lr=3e-3, weight_decay=1e-2)
    ref.policy_net_1.load_state_dict(tr.policy_net_1.state_dict())
    net, rnet = tr.policy_net_1, ref.policy_net_1
    total = sum(p.numel() for p in net.parameters())
    unused = {id(p) for p in net.unused_parameters()}
    w_unused = [p.detach().clone() for p in net.parameters() if id(p) in unused]
    g = torch.Generator(device="cpu").manual_seed(1)
    for it in range(4):
        flat = torch.randn(total, generator=g).to(dev) * 0.1
        if it < 3:
            tr._adam_step_device(0, flat)
        else:               # the fourth step by torch itself, on the moments the kernel left in its state
            net.set_flat_gradients(flat)
            tr.opts[0].step()
        rnet.set_flat_gradients(flat)
        ref.opts[0].step()
        ref.scheds[0].step()
        torch.cuda.synchronize()
        for (name, p), q in zip(net.named_parameters(), rnet.parameters()):
            assert float((p - q).detach().abs().max()) <= 2e-6 * max(1.0, float(q.detach().abs().max())), (it, name)
    for p, w in zip([p for p in net.parameters() if id(p) in unused], w_unused):
        assert torch.equal(p, w) and p not in tr.opts[0].state
    sd, rsd = tr.opts[0].state_dict(), ref.opts[0].state_dict()
    assert sorted(sd["state"]) == sorted(rsd["state"])
    for key in sd["state"]:
        assert float(sd["state"][key]["step"]) == float(rsd["state"][key]["step"]) == 4.0
        assert torch.allclose(sd["state"][key]["exp_avg"], rsd["state"][key]["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(sd["state"][key]["exp_avg_sq"], rsd["state"][key]["exp_avg_sq"], rtol=1e-5, atol=1e-9)


def test_device_learning_loop_follows_the_host_loop(lib_built):
    """`train_loop_device` (no host round trip inside a step; replay, sampling, forward + backward and Adam as
    kernels) against `train_loop_vec` (host-driven loop, autograd replayed as a HIP graph) from the same seeds: same
    actions, rewards and terminations, losses within the fp32 tolerance of the two backward implementations, and
    networks that stay together."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_device, train_loop_vec
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = _cfg()
    base = Env2DAirfoil(cfg)
    outs, trainers = [], []
    for loop in (train_loop_vec, train_loop_device):
        np.random.seed(11)
        random.seed(11)
        tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(), batch_size=8, lr=1e-3)
        venv = VecEnv2DAirfoil(cfg, 6, base_env=base, nthreads=2)
        kw = dict(chunk=4) if loop is train_loop_device else {}
        outs.append(loop(tr, venv, 9, eps_decay=2, **kw))
        trainers.append(tr)
    a, b = outs
    assert np.array_equal(a["dones"], b["dones"])
    assert np.allclose(a["rewards"], b["rewards"], rtol=1e-9, atol=1e-12)
    assert len(a["losses"]) == len(b["losses"]) == 7 and np.isfinite(b["losses"]).all()
    assert np.allclose(a["losses"], b["losses"], rtol=2e-3, atol=1e-6), (a["losses"], b["losses"])
    assert np.array_equal(a["steps_done"], b["steps_done"])
    for (name, p), q in zip(trainers[0].policy_net_1.named_parameters(), trainers[1].policy_net_1.parameters()):
        assert float((p - q).detach().abs().max()) < 2e-4 * max(1e-2, float(p.detach().abs().max())), name
    assert trainers[0].num_grads == trainers[1].num_grads and trainers[0].select == trainers[1].select


@pytest.mark.parametrize("target_update,optim_per_step", [(1, 1), (2, 1), (1, 2)])
def test_device_loop_trains_the_acting_network_on_the_side_stream(lib_built, target_update, optim_per_step):
    """The double-DQN toggle flipping every 1 / 2 gradient applications: policy_net_1 - the network the env step acts
    with on the main stream - is the one the optimiser chain of the side stream writes.  The acting copy of its packed
    parameters must follow every update exactly once, at a point ordered against `mdq_adam_step` (it was repacked by
    the acting forward while the Adam kernel was in flight, and stayed one update stale).  Against the host loop from
    the same seeds: same actions / rewards / terminations (the greedy actions see the same weights), same losses and
    the same two networks; afterwards both packed copies equal the parameters."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_device, train_loop_vec
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = _cfg()
    base = Env2DAirfoil(cfg)
    outs, trainers = [], []
    for loop in (train_loop_vec, train_loop_device):
        np.random.seed(23)
        random.seed(23)
        tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(), batch_size=8, lr=1e-3, target_update=target_update)
        venv = VecEnv2DAirfoil(cfg, 6, base_env=base, nthreads=2)
        kw = dict(chunk=5) if loop is train_loop_device else {}
        # eps_end = 0.6: a good share of greedy actions, which depend on the acting weights
        outs.append(loop(tr, venv, 12, eps_decay=2, eps_end=0.6, optim_per_step=optim_per_step, **kw))
        trainers.append(tr)
    a, b = outs
    assert np.array_equal(a["dones"], b["dones"])
    assert np.allclose(a["rewards"], b["rewards"], rtol=1e-9, atol=1e-12)
    assert len(a["losses"]) == len(b["losses"]) == 10 * optim_per_step and np.isfinite(b["losses"]).all()
    assert np.allclose(a["losses"], b["losses"], rtol=2e-3, atol=1e-6), (a["losses"], b["losses"])
    for n1, n2 in ((trainers[0].policy_net_1, trainers[1].policy_net_1), (trainers[0].policy_net_2, trainers[1].policy_net_2)):
        for (name, p), q in zip(n1.named_parameters(), n2.parameters()):
            assert float((p - q).detach().abs().max()) < 2e-4 * max(1e-2, float(p.detach().abs().max())), name
    assert trainers[0].num_grads == trainers[1].num_grads and trainers[0].select == trainers[1].select
    # the packed copies of the device trainer: a repack now must not change anything the kernels would read
    tr = trainers[1]
    torch.cuda.synchronize()
    for net in (tr.policy_net_1, tr.policy_net_2):
        for role in ("act", "train"):
            f = tr._fused_of(net, role)
            if f.desc is None:
                continue
            f._pack()
            torch.cuda.synchronize()
            after = [t.clone() for t in f._keep]
            f._version = None
            f._pack()
            torch.cuda.synchronize()
            assert all(torch.equal(x, y) for x, y in zip(after, f._keep))
            w = net.lin3.weight.detach()
            assert torch.equal(after[-2].reshape(w.shape[1], w.shape[0]), w.t().contiguous())   # lin3_w, packed [in][out]


def test_device_loop_record_ring_wraps_and_continues(lib_built):
    """A record ring much smaller than the run (5 groups of B records): the two halves of `mdq_replay_step` land in the
    right records across the wrap-around and across two calls of the loop (the second continues the ring): every
    record carries the action / reward / done of the step that wrote it last, and the next state of a non-terminal
    transition is the state its environment's next transition starts from."""
    from meshdqn_amd.env import Env2DAirfoil
    from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_device
    from meshdqn_amd.vec_env import VecEnv2DAirfoil
    cfg = _cfg()
    np.random.seed(5)
    random.seed(5)
    B, G = 6, 5
    tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(), batch_size=8, lr=1e-3, replay_capacity=B * G)
    venv = VecEnv2DAirfoil(cfg, B, base_env=Env2DAirfoil(cfg), nthreads=2)
    outs = [train_loop_device(tr, venv, 7, eps_decay=2, chunk=3)]
    rep = tr.device_memory
    assert rep.capacity == B * G and rep.steps_pushed == 7 and rep.size() == B * G
    assert len(outs[0]["losses"]) == 5 and np.isfinite(outs[0]["losses"]).all()      # steps 2..6 (12 >= 8 records from step 2)

    def check(acts, rews, dones, t_last):
        R = rep.R.cpu().numpy()
        nf, EM = rep.N * rep.F, rep.e_max
        off = 2 * nf + 4 * EM
        for t in range(t_last - G + 1, t_last + 1):             # the G steps whose records are in the ring
            rows = R[(t % G) * B:(t % G + 1) * B]
            assert np.array_equal(rows[:, off + 2], acts[t].astype(np.float32)), t
            assert np.allclose(rows[:, off + 3], rews[t].astype(np.float32), rtol=0, atol=0) and np.array_equal(rows[:, off + 4] > 0.5, dones[t]), t
            assert (rows[dones[t], nf:2 * nf] == 0).all() and (rows[dones[t], off + 1] == 0).all()
            if t < t_last:                                       # s' of step t = s of step t + 1 of the same environment
                nxt = R[((t + 1) % G) * B:((t + 1) % G + 1) * B]
                live = ~dones[t]
                assert np.array_equal(rows[live, nf:2 * nf], nxt[live, :nf]) and np.array_equal(rows[live, off + 1], nxt[live, off])
                assert np.array_equal(rows[live][:, 2 * nf + 2 * EM:off], nxt[live][:, 2 * nf:2 * nf + 2 * EM])
    check(outs[0]["actions"], outs[0]["rewards"], outs[0]["dones"], 6)
    outs.append(train_loop_device(tr, venv, 4, eps_decay=2, chunk=3))                # continues the ring: steps 7..10
    assert rep.steps_pushed == 11
    acts = np.concatenate([o["actions"] for o in outs])
    rews = np.concatenate([o["rewards"] for o in outs])
    dones = np.concatenate([o["dones"] for o in outs])
    check(acts, rews, dones, 10)
    assert len(outs[1]["losses"]) == 5 + 4 and np.isfinite(outs[1]["losses"]).all()  # every step of the second call optimises


def test_checkpoint_restart_continues_the_device_adam_state(lib_built, tmp_path):
    """`DQNTrainer.save` / `.load` (the reference's RESTART, airfoil_dqn.py:163-179) around the kernel optimiser: the
    moments and step counts written by mdq_adam_step travel through the torch optimiser's `state_dict`, and a restarted
    trainer continues with bit-identical updates."""
    from meshdqn_amd.trainer import DistContext, DQNTrainer
    dev = torch.device("cuda")
    mk = lambda: DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(device=dev), lr=2e-3, weight_decay=1e-3)   # noqa: E731
    a = mk()
    total = sum(p.numel() for p in a.policy_net_1.parameters())
    g = torch.Generator(device="cpu").manual_seed(4)
    flats = [torch.randn(total, generator=g).to(dev) * 0.1 for _ in range(5)]
    for f in flats[:3]:
        a._adam_step_device(0, f)
        a._adam_step_device(1, f * 0.5)
    a.num_grads, a.select = 6, False
    a.save(str(tmp_path), "restart_", extra=dict(steps_done=np.arange(4)))
    b = mk()
    extra = b.load(str(tmp_path), "restart_")
    assert b.num_grads == 6 and b.select is False and np.array_equal(extra["steps_done"], np.arange(4))
    for f in flats[3:]:
        for t in (a, b):
            t._adam_step_device(0, f)
            t._adam_step_device(1, f * 0.5)
    torch.cuda.synchronize()
    for na, nb in ((a.policy_net_1, b.policy_net_1), (a.policy_net_2, b.policy_net_2)):
        for (name, p), q in zip(na.named_parameters(), nb.parameters()):
            assert torch.equal(p, q), name
    sa, sb = a.opts[0].state_dict()["state"], b.opts[0].state_dict()["state"]
    assert sorted(sa) == sorted(sb) and all(float(sa[k]["step"]) == float(sb[k]["step"]) == 5.0 for k in sa)
    assert all(torch.equal(sa[k]["exp_avg"], sb[k]["exp_avg"]) and torch.equal(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"]) for k in sa)
