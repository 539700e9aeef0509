"""Device-resident rollouts (`rollout_device`: the bench / learning-loop path) of B environments: env-steps/s of the S1
(flow 0) or S3 (flow 1: one IPCS step per coarsened mesh on the flow stream) step, median of a few repeats.  Also the
command whose kernel trace tools/timeline_step.py turns into the timeline of one step:
   python tools/time_rollout.py [B] [flow] [steps] [repeats] [mesh]"""
import os, sys, time
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(1)
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
FLOW = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 50
REP = int(sys.argv[4]) if len(sys.argv) > 4 else 5
MESH = sys.argv[5] if len(sys.argv) > 5 else "ys930"
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, f"{MESH}.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True, reproducible=False, rtol=1e-10)),   # (ground truth only)
           agent_params=dict(solver_steps=int(os.environ.get("MDQ_TOOL_SOLVER_STEPS", "5000")), episodes=10, timesteps=10000, threshold=0.001,
                             N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1, time_reward=0.005,
                             save_steps=int(os.environ.get("MDQ_TOOL_SOLVER_STEPS", "5000")) // 5, goal_vertices=0.95, plot_dir=""))
base = Env2DAirfoil(cfg)
venv = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=FLOW, flow_overlap=bool(FLOW))
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda(); fused = FusedGcn(net)
venv.get_state()
rng = np.random.default_rng(1370)
def run(k):
    ex = np.array([rng.random(B) < 0.5 for _ in range(k)])
    ra = np.array([rng.integers(0, 181, B) for _ in range(k)])
    return venv.rollout_device(fused, k, ex, ra)
run(30)
rates = []
for r in range(REP):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = run(K)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rates.append(B * K / dt)
print(f"B={B} flow={FLOW} {MESH}: median {np.median(rates):.0f} env-steps/s ({1e3 * B / np.median(rates):.3f} ms per batched step; "
      f"min {min(rates):.0f} max {max(rates):.0f}); dones {int(out['dones'].sum())} codes {np.unique(out['codes']).tolist()}")
