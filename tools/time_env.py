"""Time the reference-semantics env step (S1) of the product Env2DAirfoil (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.env import Env2DAirfoil
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=int(sys.argv[1]) if len(sys.argv) > 1 else 200, episodes=10, timesteps=10000, threshold=0.001, N_closest=180,
                             gt_drag=-1, gt_time=-1, u=-1, p=-1, time_reward=0.005, save_steps=(int(sys.argv[1]) if len(sys.argv) > 1 else 200) // 5, goal_vertices=0.95, plot_dir=""))
t0 = time.time(); env = Env2DAirfoil(cfg); torch.cuda.synchronize(); print("init+reset s", time.time() - t0)
rng = np.random.default_rng(1370)
env.get_state()
import cProfile, pstats
pr = cProfile.Profile(); n = 30
t0 = time.time(); pr.enable()
for k in range(n):
    st, r, done, _ = env.step(int(rng.integers(0, 180)))
pr.disable(); dt = time.time() - t0
print(f"S1 env.step: {dt/n*1e3:.1f} ms/step  ({n/dt:.1f} steps/s, single env, host-orchestrated)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
