"""Host-side cost of the pieces of one batched env step with the GPU idle at every measurement point (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=200, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=40, goal_vertices=0.95, plot_dir=""))
venv = VecEnv2DAirfoil(cfg, 128)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); fused = FusedGcn(net.cuda())
rng = np.random.default_rng(0)
st = venv.get_state()
T = dict(act=0.0, begin=0.0, end_total=0.0, collect=0.0, restore=0.0, get_state=0.0)
def wrap(name, key):
    f = getattr(venv, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[key] += time.perf_counter() - t; return r
    setattr(venv, name, g)
wrap("_refresh_collect", "collect"); wrap("_restore_initial", "restore"); wrap("get_state", "get_state")
n = 30
for k in range(n + 5):
    if k == 5:
        for key in T: T[key] = 0.0
    torch.cuda.synchronize(); t = time.perf_counter()
    q = fused.forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], venv.N, venv.EMAX)
    torch.cuda.synchronize(); t1 = time.perf_counter()       # (includes the GPU time of the forward: reported separately below)
    acts = np.where(rng.random(128) < 0.5, rng.integers(0, 181, 128), q.argmax(1).cpu().numpy())
    T["act"] += time.perf_counter() - t1
    t = time.perf_counter(); venv.step_begin(acts); T["begin"] += time.perf_counter() - t
    torch.cuda.synchronize()
    t = time.perf_counter(); st, rew, done, info = venv.step_end(); T["end_total"] += time.perf_counter() - t
for key, v in T.items():
    print(f"{key:10s} {1e6 * v / n:8.1f} us per step (host)")
