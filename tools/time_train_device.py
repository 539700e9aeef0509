"""Time train_loop_device (S3 env step + replay + optimiser chain) with the optimiser chain on the flow stream / on a stream
of its own (dev tool).   python tools/time_train_device.py [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_device
from meshdqn_amd.vec_env import VecEnv2DAirfoil
G_ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G_, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
base = Env2DAirfoil(cfg)
for mode in ("flow", "own", "flow", "own"):
    trainer = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext())
    venv = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=1, flow_overlap=True)
    train_loop_device(trainer, venv, 6, optimiser_stream=mode)
    torch.cuda.synchronize(); t0 = time.time()
    train_loop_device(trainer, venv, n, optimiser_stream=mode)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"optimiser_stream={mode}: {dt / n * 1e3:.3f} ms per batched step -> {B * n / dt:.0f} env-steps/s; calibration flow "
          f"{[round(v, 2) for v in getattr(venv, 'calibration_ms', [])]} opt {[round(v, 2) for v in getattr(trainer, 'opt_calibration_ms', [])]}", flush=True)
    venv.flow_wait()
    del venv, trainer
