"""Distribution of the blocked smoothing kernel's repair work per launch in an S1 rollout (one launch = 128 meshes; the launch
lasts as long as its slowest mesh): repaired sweeps / repair rounds / pipelined sweeps sent back, per launch the maximum over
the meshes and the mean, and the launch duration (HIP events)."""
import os, sys, time
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(1)
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnvGroups
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
from meshdqn_amd.mesh_ops import smooth_fast_stats
B = 128
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
base = Env2DAirfoil(cfg)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda()
grp = VecEnvGroups(cfg, B, 1, base_env=base, flow_steps=0)
env = grp.envs[0]
fused = FusedGcn(net)
rng = np.random.default_rng(1370)
def run(k):
    grp.rollout_device([fused], k, [np.array([rng.random(B) < 0.5 for _ in range(k)])], [np.array([rng.integers(0, 181, B) for _ in range(k)])])
run(10)
rows = []
for step in range(K):
    env.smooth_events = []
    run(1)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in env.smooth_events]
    st = smooth_fast_stats(env.device, B, env.NV)
    rows.append((ms[0] if ms else float("nan"), st))
env.smooth_events = None
print("launch ms | max/mean repaired sweeps | max/mean repair rounds | max/mean sent back | handed back")
for ms, st in rows:
    sb = st[:, 3] & 0xFFFF
    print(f"{ms:7.3f} | {st[:,1].max():2d} {st[:,1].mean():5.2f} | {st[:,2].max():2d} {st[:,2].mean():5.2f} | {sb.max():2d} {sb.mean():5.2f} | {int((st[:,0] != 0).sum())}")
a = np.array([r[0] for r in rows]); mr = np.array([r[1][:, 2].max() for r in rows]); ms_ = np.array([(r[1][:, 3] & 0xFFFF).max() for r in rows])
print("mean launch", a.mean(), "corr with max rounds", np.corrcoef(a, mr)[0, 1], "with max sentback", np.corrcoef(a, ms_)[0, 1])
for v in sorted(set(mr)):
    print(f"max rounds {v}: {np.sum(mr == v)} launches, mean {a[mr == v].mean():.3f} ms")
