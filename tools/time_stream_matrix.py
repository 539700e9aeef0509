import os, sys, time
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.set_num_threads(1)
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
G = "/root/repo/tests/golden"
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
base = Env2DAirfoil(cfg)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda()
rng = np.random.default_rng(1370)
B, K = 128, 30
env = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=1, flow_overlap=True, flow_pressure="direct")
fg = FusedGcn(net)
def run(k, st):
    with torch.cuda.stream(st):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        env.rollout_device(fg, k, rng.random((k, B)) < 0.5, rng.integers(0, 181, (k, B)))
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
from meshdqn_amd.streams import _overlaps
mains = [torch.cuda.Stream() for _ in range(6)]
flows = [env._flow_stream] + [torch.cuda.Stream() for _ in range(7)]
run(10, mains[0])
print("probe (1 = streams.py says they overlap):")
for mi, m in enumerate(mains):
    print(mi, " ".join(f"{int(_overlaps(f, m, torch.device('cuda'))):5d}" for f in flows), flush=True)
print("rows: main stream 0..5; columns: flow stream 0..7 (ms per batched step)")
for mi, m in enumerate(mains):
    row = []
    for f in flows:
        env.flow_wait(); env._flow_stream = f
        run(4, m)
        row.append(run(K, m))
    print(mi, " ".join(f"{t:5.2f}" for t in row), flush=True)
