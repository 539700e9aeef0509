"""Time the batched reference-semantics env step (S1) and S1 + Q-forward (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
FLOW = int(sys.argv[2]) if len(sys.argv) > 2 else 0
GSM = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
OVL = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
print("host cores", os.cpu_count())
t0 = time.time(); venv = VecEnv2DAirfoil(cfg, B, flow_steps=FLOW, gpu_smoothing=GSM, flow_overlap=OVL); print("init s", time.time() - t0)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda(); fused = FusedGcn(net)
st = venv.get_state()
rng = np.random.default_rng(0)
import cProfile, pstats
for rep in range(2):
    pr = cProfile.Profile()
    n = 10; torch.cuda.synchronize(); t0 = time.time()
    if rep: pr.enable()
    for k in range(n):
        q = fused.forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], venv.N, venv.EMAX)
        greedy = q.argmax(1).cpu().numpy()
        acts = np.where(rng.random(B) < 0.5, rng.integers(0, 181, B), greedy)
        st, rew, done, info = venv.step(acts)
    torch.cuda.synchronize(); dt = time.time() - t0
    if rep: pr.disable()
    print(f"B={B}: {dt/n*1e3:.1f} ms per batched S1 step + Q-forward -> {B*n/dt:.0f} env-steps/s")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
