"""Deployment evaluator: counterpart of the reference's `deploy_dqn.py` (greedy rollout + full re-simulation after every
removal, deploy_dqn.py:262-269, :299-463, :495-517).

    out = deploy(env, net, complete_traj=True, save_dir=...)

Reference flow (SURROGATE_MODEL False, MULTI_SNAPSHOT True, complete_traj True: the stock flags, deploy_dqn.py:20-21,58):
roll the policy (or replay a recorded action list, the 'best episode' replay `:324-331`); after EVERY selected removal
re-simulate `solver_steps` IPCS steps from rest on the coarsened mesh (`run_sim`, `:262-269,376-387`); keep the mesh of
the last step as `best_mesh` (`:427-431`), put it back (`:440-441`), re-mesh once more and run the final simulation whose
last drag is compared with the ground truth (`:495-517`).  Output arrays in the reference's layouts:
  interpolate_drag_trajectory  rows [nv, S drags, S lifts]: the initial mesh, then one row per step  (`:304-313,364-366,396`)
  drag_trajectory              rows [nv, S drags, S lifts]: [nv0, gt_drag, gt_lift], then one row per re-simulation (`:303,378-383,410-413`)
  complete_drags / _lifts      [gt_drag] + the S snapshot drags of every re-simulation  (`:315-316,384-387,458-463`)
  actions                      (ours) the actions taken

MI355X shape of the work (`batched=True`, default): the re-simulations do not feed the action selection (the next step
re-meshes and restarts the flow from rest, flow_solver.py:233-359), so the policy is rolled FIRST - env steps only - and
all M coarsened meshes of the episode (+ the mesh of the final simulation) are re-simulated together as ONE `IpcsBatch`:
M environments x `solver_steps` steps in `solver_steps` launches instead of M x `solver_steps` (one workgroup per mesh:
44 meshes use 44 CUs instead of 1).  `batched=False` keeps the reference's order (DEPLOY mode: every step re-assembles and
re-factorises, then `run_sim`); both give the same files BIT FOR BIT at the stock 5000 steps (the flow solver's default is
the reproducible operator mode 2: one workgroup per mesh, fixed summation order, so a mesh's trajectory does not depend
on the batch it is simulated in; tests/test_deploy_gpu.py).
"""
from __future__ import annotations

import os
import time
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


def resimulate_batch(env, meshes):
    """`run_sim` for a list of (coords, cells) meshes at once: one `IpcsBatch` (device assembly, device pressure
    factorisation for solver_type 'lu'), `solver_steps` steps from rest, forces every `save_steps` steps.
    Returns (drags (M, S), lifts (M, S))."""
    from .ipcs_batch import IpcsBatch
    from .topology import MeshTopology
    fs = env.flow_solver
    topos = [MeshTopology(np.asarray(c, np.float64), np.asarray(t)) for c, t in meshes]
    batch = IpcsBatch(topos, [t.coords for t in topos], mu=fs.mu, rho=fs.rho, dt=fs.dt_value, rtol=fs.rtol,
                      device=fs.device, mode=getattr(fs, "mode", -1),
                      pressure_direct=("device" if fs.solver_type == "lu" else False))
    batch.assemble()
    drags, lifts, done = [], [], 0
    while done < env.solver_steps:
        n = min(env.save_steps - (done % env.save_steps), env.solver_steps - done)
        d, l = batch.evolve(n)
        done += n
        if done % env.save_steps == 0:
            drags.append(d[:, -1].clone())
            lifts.append(l[:, -1].clone())
    out = torch.stack(drags, 1).cpu().numpy(), torch.stack(lifts, 1).cpu().numpy()
    batch.check()               # (an abandoned step - team barrier time-out - is an error, not a NaN row in the evaluation)
    return out


@torch.no_grad()
def select_action(net, state, device):
    """Greedy action of the trained network (deploy_dqn.py:201-202)."""
    st = state.to(device)
    q = net.forward_fused(st) if device.type == "cuda" else net(st)
    return int(q.argmax().item())


def deploy(env, net=None, actions=None, complete_traj: bool = True, max_steps: Optional[int] = None,
           save_dir: Optional[str] = None, prefix: str = "", stop_on_done: bool = True, batched: bool = True,
           final_sim: bool = True):
    """Roll the policy (or a recorded action list) until the environment terminates; returns a dict of trajectories and
    writes the reference's `.npy` files into `save_dir` (names `{prefix}interpolate_drag_trajectory.npy`,
    `{prefix}drag_trajectory.npy`, `{prefix}complete_drags.npy`, `{prefix}complete_lifts.npy`, `{prefix}actions.npy`)."""
    from .flow_solver import Mesh
    dev = env.compute_device
    fs = env.flow_solver
    if not batched:
        fs.deploy()                                  # deploy_dqn.py:86: every remesh re-assembles + re-factorises
    state = env.get_state()
    _ = env.calculate_reward()
    nv0 = len(fs.mesh.coordinates())
    gt_drag, gt_lift = np.array(env.gt_drag, np.float64), np.array(env.gt_lift, np.float64)
    est_v, est_d, est_l = [nv0], [np.array(env.new_drags)], [np.array(env.new_lifts)]           # :304-313
    traj_v, traj_d, traj_l = [nv0], [gt_drag], [gt_lift]                                          # :303
    complete_d, complete_l = [gt_drag], [gt_lift]                                                 # :315-316
    taken, selected_ids, pending = [], [], []
    best_mesh = Mesh(fs.mesh)
    t, done = 0, False
    t_roll = time.perf_counter()
    while True:
        if actions is not None:
            if t >= len(actions):                   # (the replayed episode is over, :328-331)
                break
            a = int(actions[t])
        else:
            a = select_action(net, state, dev)
        selected = env.coord_map.get(a, None)        # KeyError in the reference = "NO REMOVAL" (:355-358)
        state, reward, done, _ = env.step(a)
        taken.append(a)
        selected_ids.append(np.nan if selected is None else int(selected))
        nv_after = len(fs.mesh.coordinates())
        est_v.append(nv_after)
        est_d.append(np.array(env.new_drags))
        est_l.append(np.array(env.new_lifts))
        if complete_traj and selected is not None:   # :376-387
            traj_v.append(nv_after)
            if batched:
                pending.append((fs.mesh.coordinates().copy(), fs.mesh.cells().copy()))
            else:
                d, l, full_d, full_l = run_sim(env)
                traj_d.append(np.array(full_d))
                traj_l.append(np.array(full_l))
                complete_d.append(np.array(full_d))
                complete_l.append(np.array(full_l))
        best_mesh = Mesh(fs.mesh)                    # :427-431 (assigned before the `done` test as well)
        t += 1
        if (done and stop_on_done) or (max_steps is not None and t >= max_steps):
            break
    t_roll = time.perf_counter() - t_roll
    # the mesh of the final simulation: best_mesh put back and re-meshed once more (smooth(50) again, :440-441,:498)
    final_mesh = None
    if final_sim:
        final_mesh = Mesh(best_mesh)
        if fs.smooth:
            final_mesh.smooth(50)
    t_sim = time.perf_counter()
    final = None
    if batched:
        todo = list(pending) + ([(final_mesh.coordinates(), final_mesh.cells())] if final_sim else [])
        if todo:
            D, L = resimulate_batch(env, todo)
            for k in range(len(pending)):
                traj_d.append(D[k]); traj_l.append(L[k]); complete_d.append(D[k]); complete_l.append(L[k])
            if final_sim:
                final = float(D[-1, -1])
    elif final_sim:
        fs.mesh = Mesh(best_mesh)                    # "PUTTING MESH BACK"
        fs.remesh(Mesh(best_mesh))
        _, _, full_d, _ = run_sim(env)
        final = float(full_d[-1])
    t_sim = time.perf_counter() - t_sim
    if final_sim:                                    # the environment ends on the last acceptable mesh either way
        fs.mesh = Mesh(best_mesh)
    out = dict(actions=np.array(taken), selected=np.array(selected_ids, dtype=np.float64),
               est_vertices=np.array(est_v), est_drag=np.array(est_d), est_lift=np.array(est_l),
               traj_vertices=np.array(traj_v), traj_drag=np.array(traj_d), traj_lift=np.array(traj_l),
               complete_drags=np.array(complete_d), complete_lifts=np.array(complete_l),
               gt_drag=gt_drag, gt_lift=gt_lift, done=bool(done), rollout_seconds=t_roll, resimulation_seconds=t_sim,
               resimulated_meshes=len(traj_v) - 1 + (1 if final_sim else 0), batched=bool(batched))
    out["interpolate_drag_trajectory"] = np.hstack((out["est_vertices"][:, None], out["est_drag"], out["est_lift"]))
    if complete_traj:
        out["drag_trajectory"] = np.hstack((out["traj_vertices"][:, None], out["traj_drag"], out["traj_lift"]))
    if final is not None:                            # :505-517
        out["new_drag"] = final
        out["final_vertices"] = len(best_mesh.coordinates())
        out["drag_error_percent"] = float(100.0 * abs(final - gt_drag[-1]) / abs(gt_drag[-1]))
        out["final_drag_error"] = out["drag_error_percent"] / 100.0
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        names = ["interpolate_drag_trajectory"] + (["drag_trajectory", "complete_drags", "complete_lifts"] if complete_traj else [])
        for k in names:
            np.save(os.path.join(save_dir, f"{prefix}{k}.npy"), out[k])
        np.save(os.path.join(save_dir, f"{prefix}actions.npy"), out["actions"])
    return out
