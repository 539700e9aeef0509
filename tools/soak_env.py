"""Soak test of the device-resident batched env (dev tool): many steps, random + greedy actions, invariants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
G_ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
mesh = sys.argv[2] if len(sys.argv) > 2 else "ys930"
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G_, mesh + ".npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=5000, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=1000, goal_vertices=0.95, plot_dir=""))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
venv = VecEnv2DAirfoil(cfg, 128, flow_steps=1)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); fused = FusedGcn(net.cuda())
rng = np.random.default_rng(0)
st = venv.get_state(); ndone = 0; minnv = 10**9; codes = np.zeros(3, int)
t0 = time.time()
for k in range(steps):
    q = fused.forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], venv.N, venv.EMAX)
    assert torch.isfinite(q).all()
    acts = np.where(rng.random(128) < 0.7, rng.integers(0, 181, 128), q.argmax(1).cpu().numpy())
    st, rew, done, info = venv.step(acts)
    assert np.isfinite(rew).all() and np.isfinite(info["new_drags"]).all() and np.isfinite(info["flow_drag"]).all()
    assert (info["nv"] <= venv.NV).all() and (info["nv"] >= 0.94 * venv.NV - 1).all()
    assert (st["nedges"] % 3 == 0).all() and (st["nsel"] <= 180).all()
    ndone += int(done.sum()); minnv = min(minnv, int(info["nv"].min()))
    for c in (0, 1, 2): codes[c] += int((info["code"] == c).sum())
print(f"{mesh}: {steps} batched steps ok in {time.time()-t0:.1f} s; episodes finished {ndone}; min nv {minnv}; codes {codes.tolist()}; "
      f"mean |flow drag| {np.abs(info['flow_drag']).mean():.5f}")
