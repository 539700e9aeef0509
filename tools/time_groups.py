"""Time S1 / S3 batched env steps with G concurrently stepped groups (dev tool): B G flow_steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnvGroups
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
sys.setswitchinterval(float(os.environ.get("SWI", "0.005")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
FLOW = int(sys.argv[2]) if len(sys.argv) > 2 else 0
G_ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G_, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
base = Env2DAirfoil(cfg)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda()
for G in (1, 2, 4, 8):
    groups = VecEnvGroups(cfg, B, G, base_env=base, flow_steps=FLOW)
    fused = [FusedGcn(net) for _ in range(G)]
    rngs = [np.random.default_rng(g) for g in range(G)]
    def act(g, env, st):
        q = fused[g].forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], env.N, env.EMAX)
        greedy = q.argmax(1).cpu().numpy()
        return np.where(rngs[g].random(env.B) < 0.5, rngs[g].integers(0, 181, env.B), greedy)
    groups.rollout(act, 3)
    n = 20; torch.cuda.synchronize(); t0 = time.time()
    groups.rollout(act, n)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"B={B} G={G} flow={FLOW}: {dt/n*1e3:.2f} ms per batched step -> {B*n/dt:.0f} env-steps/s", flush=True)
