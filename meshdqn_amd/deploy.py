"""Deployment evaluator: counterpart of the reference's `deploy_dqn.py` (greedy rollout + full
re-simulation after every removal, deploy_dqn.py:262-269, :318-424, :495-517).

    out = deploy(env, net, complete_traj=True, save_dir=...)

The environment is switched to DEPLOY mode (`env.flow_solver.deploy()`, deploy_dqn.py:86): every
accepted removal re-assembles the three IPCS operators on the coarsened mesh (HIP assembly kernel +
host pressure factorisation) and `run_sim` advances `solver_steps` IPCS steps from rest in chunks of
`save_steps` steps per kernel launch.  Output arrays use the reference's layouts:
  interpolate_drag_trajectory  rows [nv, S drags, S lifts]  (from the interpolated snapshots, :396)
  drag_trajectory              rows [nv, S drags, S lifts]  (from the full simulations,      :410-413)
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch


def run_sim(env):
    """deploy_dqn.py:262-269: solver_steps x evolve() on the current (re-assembled) mesh."""
    drags, lifts = [], []
    done = 0
    fs = env.flow_solver
    while done < env.solver_steps:
        n = min(env.save_steps - (done % env.save_steps), env.solver_steps - done)
        u, p, drag, lift = fs.evolve(n)
        done += n
        if done % env.save_steps == 0:
            drags.append(drag)
            lifts.append(lift)
    return drags[-1], lifts[-1], drags, lifts


@torch.no_grad()
def select_action(net, state, device):
    """Greedy action of the trained network (deploy_dqn.py:201-202)."""
    st = state.to(device)
    q = net.forward_fused(st) if device.type == "cuda" else net(st)
    return int(q.argmax().item())


def deploy(env, net=None, actions=None, complete_traj: bool = True, max_steps: Optional[int] = None,
           save_dir: Optional[str] = None, prefix: str = "", stop_on_done: bool = True):
    """Roll the policy (or a recorded action list, the reference's 'best episode' replay) until the
    environment terminates; returns a dict of trajectories."""
    dev = env.compute_device
    env.flow_solver.deploy()
    state = env.get_state()
    _ = env.calculate_reward()
    est_v, est_d, est_l = [], [], []
    traj_v, traj_d, traj_l, taken = [], [], [], []
    t = 0
    while True:
        if actions is not None:
            if t >= len(actions):
                break
            a = int(actions[t])
        else:
            a = select_action(net, state, dev)
        nv_before = len(env.flow_solver.mesh.coordinates())
        state, reward, done, _ = env.step(a)
        taken.append(a)
        nv_after = len(env.flow_solver.mesh.coordinates())
        removed = nv_after < nv_before
        if hasattr(env, "new_drags"):
            est_v.append(nv_after)
            est_d.append(np.array(env.new_drags))
            est_l.append(np.array(env.new_lifts))
        if complete_traj and removed:
            d, l, full_d, full_l = run_sim(env)
            traj_v.append(nv_after)
            traj_d.append(full_d)
            traj_l.append(full_l)
        t += 1
        if (done and stop_on_done) or (max_steps is not None and t >= max_steps):
            break
    out = dict(actions=np.array(taken), est_vertices=np.array(est_v), est_drag=np.array(est_d), est_lift=np.array(est_l),
               traj_vertices=np.array(traj_v), traj_drag=np.array(traj_d), traj_lift=np.array(traj_l),
               gt_drag=np.array(env.gt_drag), gt_lift=np.array(env.gt_lift))
    if len(est_v):
        out["interpolate_drag_trajectory"] = np.hstack((out["est_vertices"][:, None], out["est_drag"], out["est_lift"]))
    if len(traj_v):
        out["drag_trajectory"] = np.hstack((out["traj_vertices"][:, None], out["traj_drag"], out["traj_lift"]))
        out["final_drag_error"] = float(abs(out["traj_drag"][-1][-1] - env.gt_drag[-1]) / abs(env.gt_drag[-1]))
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        for k in ("interpolate_drag_trajectory", "drag_trajectory"):
            if k in out:
                np.save(os.path.join(save_dir, f"{prefix}{k}.npy"), out[k])
        np.save(os.path.join(save_dir, f"{prefix}actions.npy"), out["actions"])
    return out
