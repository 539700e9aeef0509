"""Time the batched DQN loop: env step + replay push + optimise (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_vec
from meshdqn_amd.vec_env import VecEnv2DAirfoil
G_ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G_, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = DistContext()
trainer = DQNTrainer(n_actions=180, num_inputs=17, ctx=ctx)
venv = VecEnv2DAirfoil(cfg, B, base_env=Env2DAirfoil(cfg))
train_loop_vec(trainer, venv, 3)
torch.cuda.synchronize(); t0 = time.time(); n = 10
train_loop_vec(trainer, venv, n)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"B={B}: {dt/n*1e3:.1f} ms per batched training step (env step + {B} replay pushes + 1 optimise) -> {B*n/dt:.0f} env-steps/s")
trs = (trainer.device_memory or trainer.memory).sample(32)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(10): trainer.optimize(trs)
torch.cuda.synchronize(); print(f"optimise alone: {(time.time()-t0)/10*1e3:.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); train_loop_vec(trainer, venv, 5); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
