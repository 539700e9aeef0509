"""GPU: `train.py` itself as a multi-rank job ("train.py has never been started with N > 1 ranks in any test").
Eight ranks share cuda:0 over gloo (MDQ_SHARE_GPU=1 MDQ_DIST_BACKEND=gloo: RCCL refuses two ranks on one device), started by
`train.py --gpus 8` through meshdqn_amd/launcher.py - the parent never touches the GPU, nothing re-execs.  What a real
8-GPU run then needs beyond this is RCCL itself.  Reference: the parameter-server round trip of airfoil_dqn.py:412-420
(every worker applies the same gradients) and the TorchTrainer job of :508-514."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digests(save_dir, prefix, world):
    out = []
    for r in range(world):
        with open(os.path.join(save_dir, f"{prefix}digest_rank{r}.jsonl")) as f:
            out.append([json.loads(l) for l in f if l.strip()])
    return out


@pytest.mark.slow
def test_train_py_eight_ranks_share_one_gpu_and_restart(lib_built, tmp_path):
    """10 batched steps of the device-resident learning loop on 8 x 4 environments with the shared replay, a checkpoint, then
    `--restart`: after EVERY batched step both Q-networks and the Adam moments are bit-identical on all eight ranks (the flat
    gradient all-reduce really averaged and every rank took the same step), the record rings hold the same bytes (the
    all-gather delivered everybody's transitions everywhere), the parameters do move, and the restarted job continues the
    epsilon counters and the checkpointed networks of the first one."""
    W, B = 8, 4
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "ray_ys930.yaml")))
    cfg["flow_config"]["geometry_params"]["mesh"] = os.path.join(ROOT, "tests", "golden", "ys930.npz")
    cfg["agent_params"].update(solver_steps=100, save_steps=20)
    cpath = os.path.join(str(tmp_path), "cfg.yaml")
    yaml.safe_dump(cfg, open(cpath, "w"))
    save = os.path.join(str(tmp_path), "run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MDQ_SHARE_GPU="1", MDQ_DIST_BACKEND="gloo")
    base = [sys.executable, "train.py", "--gpus", str(W), "--config", cpath, "--envs", str(B), "--share-replay", "--save-dir", save,
            "--save-every", "5", "--digest"]
    out = subprocess.run(base + ["--steps", "10"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    assert f"ranks {W}: 10 batched steps x {B} envs/rank" in out.stdout
    d = _digests(save, "", W)
    assert all(len(x) == 10 and [l["step"] for l in x] == list(range(1, 11)) for x in d)
    for k in range(10):
        for key in ("net1", "net2", "optimiser", "ring", "optimiser_steps"):
            assert len({d[r][k][key] for r in range(W)}) == 1, (k, key)          # identical on every rank
    assert d[0][-1]["optimiser_steps"] >= 6                                       # 32 records per step: sampling from step 2 on
    assert d[0][0]["net1"] != d[0][-1]["net1"] or d[0][0]["net2"] != d[0][-1]["net2"]     # ... and the weights moved
    assert len({d[0][k]["ring"] for k in range(1, 10)}) == 9                      # the finished part grows by W x B records per step
    assert all(d[r][-1]["steps_done"] == [10] * B for r in range(W))
    for f in ("policy_net_1.pt", "policy_net_2.pt", "config.yaml", "step_rewards.npy"):
        assert os.path.exists(os.path.join(save, f)), f
    assert np.load(os.path.join(save, "step_rewards.npy")).shape == (10, B)
    # ---- restart: reads the checkpoint of step 10, writes with one "restart_" prefix
    out = subprocess.run(base + ["--steps", "4", "--restart"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-4000:]
    r = _digests(save, "restart_", W)
    assert all(len(x) == 4 for x in r)
    for k in range(4):
        for key in ("net1", "net2", "optimiser"):
            assert len({r[q][k][key] for q in range(W)}) == 1, (k, key)
    assert r[0][0]["steps_done"] == [11] * B and r[0][-1]["steps_done"] == [14] * B     # the epsilon counters continue
    assert os.path.exists(os.path.join(save, "restart_policy_net_1.pt"))
    # the restarted replicas start from the checkpointed networks: no optimiser step before the ring holds a minibatch again,
    # so the first digest of the restart IS the last one of the first job
    assert (r[0][0]["net1"], r[0][0]["net2"]) == (d[0][-1]["net1"], d[0][-1]["net2"])
    assert (r[0][-1]["net1"], r[0][-1]["net2"]) != (d[0][-1]["net1"], d[0][-1]["net2"])      # and learning goes on


def test_train_py_refuses_more_ranks_than_gpus(lib_built, tmp_path):
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with a single GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MDQ_SHARE_GPU")}
    out = subprocess.run([sys.executable, "train.py", "--gpus", "2", "--config", os.path.join(ROOT, "configs", "ray_ys930.yaml")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "needs 2 GPUs" in out.stderr


def test_train_py_one_rank_through_rccl_equals_no_process_group(lib_built, tmp_path):
    """The device-resident learning loop of `train.py` with its collectives going through RCCL - ONE rank, a process group of
    one (`MDQ_FORCE_COLLECTIVES=1`, backend "nccl"): the flat gradient all-reduce inside every optimiser step and the in-place
    record all-gather on the optimiser stream really run on the device's communicator; networks, Adam moments and the record
    ring after every batched step are bit-identical to the same job without a process group.  (More than one rank over RCCL
    needs more than one GPU; the eight-rank job above runs over gloo.)"""
    import socket
    B = 8
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "ray_ys930.yaml")))
    cfg["flow_config"]["geometry_params"]["mesh"] = os.path.join(ROOT, "tests", "golden", "ys930.npz")
    cfg["agent_params"].update(solver_steps=100, save_steps=20)
    cpath = os.path.join(str(tmp_path), "cfg.yaml")
    yaml.safe_dump(cfg, open(cpath, "w"))
    digs = []
    for forced in (False, True):
        save = os.path.join(str(tmp_path), "rccl" if forced else "plain")
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MDQ_SHARE_GPU", "MDQ_DIST_BACKEND")}
        if forced:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            env.update(MDQ_FORCE_COLLECTIVES="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        out = subprocess.run([sys.executable, "train.py", "--config", cpath, "--envs", str(B), "--share-replay", "--save-dir", save,
                              "--steps", "10", "--digest"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-4000:]
        if forced:
            assert "nccl" in out.stdout + out.stderr, (out.stdout[-2000:], out.stderr[-2000:])
        digs.append(_digests(save, "", 1)[0])
    assert len(digs[0]) == len(digs[1]) == 10 and digs[0][-1]["optimiser_steps"] >= 4
    for a, b in zip(*digs):
        for key in ("net1", "net2", "optimiser", "ring", "optimiser_steps", "steps_done"):
            assert a[key] == b[key], (a["step"], key)
