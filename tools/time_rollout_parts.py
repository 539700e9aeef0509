"""Where the time of a K-step device-resident rollout goes outside its steps: rollout_begin (uploads), the host enqueue,
rollout_end (read-back + host mirrors), the tail of the flow stream.  python tools/time_rollout_parts.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
torch.set_num_threads(1)
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
B = 128
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(R, "tests/golden/ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True, reproducible=False, rtol=1e-10)),
           agent_params=dict(solver_steps=200, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=40, goal_vertices=0.95, plot_dir=""))
venv = VecEnv2DAirfoil(cfg, B, flow_steps=1, flow_overlap=True)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda(); fused = FusedGcn(net)
venv.get_state()
rng = np.random.default_rng(1)
from meshdqn_amd.streams import role_streams
MAIN = role_streams(venv.device)["main"]
def once(K):
    with torch.cuda.stream(MAIN):
        return _once(K)
def _once(K):
    ex = np.array([rng.random(B) < 0.5 for _ in range(K)]); ra = np.array([rng.integers(0, 181, B) for _ in range(K)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro = venv.rollout_begin(K, ex, ra)
    t1 = time.perf_counter()
    for _ in range(K): venv.rollout_step(ro, fused)
    t2 = time.perf_counter()
    torch.cuda.current_stream().synchronize()
    t3 = time.perf_counter()
    out = venv.rollout_end(ro)
    t4 = time.perf_counter()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    return [(t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t5 - t4) * 1e6, (t5 - t0) * 1e6]
once(20); once(20)
for K in (20, 50):
    r = np.median([once(K) for _ in range(7)], axis=0)
    print(f"K={K}: begin {r[0]:.0f} us, enqueue of the steps {r[1]:.0f}, wait for the main stream {r[2]:.0f}, rollout_end {r[3]:.0f}, rest (flow stream) {r[4]:.0f}; total {r[5]:.0f} = {r[5]/K:.1f} per step")
