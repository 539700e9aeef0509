"""Soak of the S3 env step on the red-refined ys930 (device-built tile maps, Morton-sorted cells, matrix-on-chip pressure solve, two
workgroups per environment): `steps` device-resident steps of B environments in rollouts of 10, with auto-reset; every rollout
checks the flow leg (status words, finite forces: `rollout_end` / `flow_wait` raise otherwise) and prints the iteration counts, how many
environments kept the dof <- slot path (tile maps not built) and the vertex range.   python tools/soak_refined_s3.py [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(1)
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "oracle_stock_ys930_refined.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True, reproducible=False, rtol=1e-10)),
           agent_params=dict(solver_steps=50, episodes=10, timesteps=10000, threshold=10.0, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=10, goal_vertices=0.95, plot_dir=""))      # (threshold 10: episodes run to the vertex criterion, 167 removals)
base = Env2DAirfoil(cfg)
venv = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=1, flow_overlap=True)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda(); fused = FusedGcn(net)
venv.get_state()
rng = np.random.default_rng(1370)
done_total, t0, worst = 0, time.time(), 0
for r0 in range(0, STEPS, 10):
    ex = np.array([rng.random(B) < 0.5 for _ in range(10)]); ra = np.array([rng.integers(0, 181, B) for _ in range(10)])
    out = venv.rollout_device(fused, 10, ex, ra)
    fd, fl = venv.flow_wait()
    it = venv.flow_iters.cpu().numpy()
    rc = venv.flow_ts[0]["mf_rcnt"].cpu().numpy() if getattr(venv, "_flow_tile_maps", False) else None
    done_total += int(out["dones"].sum())
    worst = max(worst, int(it[:, 1].max()))
    assert np.isfinite(out["rewards"]).all() and (out["codes"] == 0).all() and np.isfinite(fd).all()
    if r0 % 50 == 0 or r0 + 10 >= STEPS:
        print(f"step {r0 + 10}: nv {int(venv.nv.min())}..{int(venv.nv.max())}, iters mean {it.mean(0).round(1).tolist()} max {it.max(0).tolist()}, "
              f"maps not built {None if rc is None else int((rc[:, 0] < 0).sum())}, rows per chunk max {None if rc is None else int(rc.max())}, "
              f"episodes ended {done_total}, {time.time() - t0:.0f}s", flush=True)
print("ok: worst pressure iterations", worst)
