"""Is the device-resident learning loop bound by the host?  Time from the start of a chunk to the moment the host has ENQUEUED all
of it (entry of `rollout_end`) against the time the GPU needs for it (return of `rollout_end`).   python tools/time_train_enqueue.py [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
torch.set_num_threads(1)
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_device
from meshdqn_amd.vec_env import VecEnv2DAirfoil
G_ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G_, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
base = Env2DAirfoil(cfg)
trainer = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext())
venv = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=1, flow_overlap=True)
train_loop_device(trainer, venv, 8)
marks = {}
orig_end, orig_begin = venv.rollout_end, venv.rollout_begin
def begin(*a, **k):
    torch.cuda.synchronize(); marks["t0"] = time.perf_counter()
    return orig_begin(*a, **k)
def end(ro):
    marks["enq"] = time.perf_counter()
    out = orig_end(ro)
    torch.cuda.synchronize(); marks["done"] = time.perf_counter()
    return out
venv.rollout_begin, venv.rollout_end = begin, end
for rep in range(3):
    train_loop_device(trainer, venv, n, chunk=n)
    print(f"chunk of {n} steps: host enqueued everything after {(marks['enq'] - marks['t0']) / n * 1e3:.3f} ms per step, the GPU finished after "
          f"{(marks['done'] - marks['t0']) / n * 1e3:.3f} ms per step", flush=True)
