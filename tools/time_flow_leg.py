"""Duration of the flow leg of the S3 step by HIP events on the flow stream (first launch of the leg's topology run ...
end of the last IPCS kernel), inside device-resident rollouts, against the step: python tools/time_flow_leg.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(1)
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True, reproducible=False, rtol=1e-10)),
           agent_params=dict(solver_steps=5000, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=1000, goal_vertices=0.95, plot_dir=""))
base = Env2DAirfoil(cfg)
venv = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=1, flow_overlap=True)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda(); fused = FusedGcn(net)
venv.get_state()
rng = np.random.default_rng(1370)
def run(k):
    ex = np.array([rng.random(B) < 0.5 for _ in range(k)]); ra = np.array([rng.integers(0, 181, B) for _ in range(k)])
    return venv.rollout_device(fused, k, ex, ra)
run(30)
resets = []
_orig_reset = venv._flow_reset
def _reset(d):
    ea = torch.cuda.Event(enable_timing=True); ea.record()
    _orig_reset(d)
    eb = torch.cuda.Event(enable_timing=True); eb.record()
    resets.append((ea, eb))
venv._flow_reset = _reset
venv.flow_events = []
torch.cuda.synchronize(); t0 = time.perf_counter()
run(50)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
legs = np.array([a.elapsed_time(b) for a, b in venv.flow_events]) * 1e3
ev = venv.flow_events
venv.flow_events = None
n = min(len(ev), len(resets))
wait = np.array([resets[k][1].elapsed_time(ev[k][0]) for k in range(5, n)]) * 1e3        # after the reset .. leg start = wait for the meshes
rst = np.array([resets[k][0].elapsed_time(resets[k][1]) for k in range(5, n)]) * 1e3
idle = np.array([ev[k - 1][1].elapsed_time(resets[k][0]) for k in range(6, n)]) * 1e3      # end of leg k - 1 .. first launch of step k
per = np.array([ev[k - 1][0].elapsed_time(ev[k][0]) for k in range(6, n)]) * 1e3
print(f"flow stream per step (us, medians): period {np.median(per):.1f} = leg {np.median(legs):.1f} + end-of-leg .. reset launch {np.median(idle):.1f} "
      f"+ reset {np.median(rst):.1f} + wait for the meshes of the step {np.median(wait):.1f}")
print(f"B={B}: step {dt / 50 * 1e6:.1f} us; flow leg (topology .. correction, events on the flow stream) median {np.median(legs):.1f} us "
      f"(min {legs.min():.1f}, max {legs.max():.1f}); the rest of the flow stream's period: history reset + waits + event records")
