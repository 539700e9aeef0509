"""Learning-loop rate (train_loop_device, S3 env step) with the Jacobi-CG pressure solve vs the per-step device
re-factorisation + direct solve, alternating in one process."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from meshdqn_amd.env import Env2DAirfoil  # noqa: E402
from meshdqn_amd.trainer import DistContext, DQNTrainer, train_loop_device  # noqa: E402
from meshdqn_amd.vec_env import VecEnv2DAirfoil  # noqa: E402


class Args:
    envs, mesh, rtol, s1_solver_steps = 128, "ys930", 1e-10, 500


dev = torch.device("cuda", 0)
cfg = bench._env_config(Args)
base = Env2DAirfoil(cfg, compute_device=dev)
for rep in range(3):
    for fp, osm in (("cg", "own"), ("cg", "flow"), ("direct", "own"), ("direct", "flow")):
        tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext(device=dev))
        venv = VecEnv2DAirfoil(cfg, 128, compute_device=dev, base_env=base, flow_steps=1, flow_overlap=True, flow_pressure=fp)
        train_loop_device(tr, venv, 4, optimiser_stream=osm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        train_loop_device(tr, venv, 40, optimiser_stream=osm)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        print(f"rep {rep} flow_pressure={fp:6s} optimiser on {osm:4s} stream: {dt * 1e3:.3f} ms per batched step ({128 / dt:.0f} env-steps/s); "
              f"calibration flow {[round(v, 2) for v in venv.calibration_ms]} optimiser {[round(v, 2) for v in getattr(tr, 'opt_calibration_ms', [])]}", flush=True)
        venv.flow_wait()
        del venv, tr
