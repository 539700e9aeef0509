"""The S3 step with the Jacobi-CG pressure solve vs the per-step device re-factorisation + direct solve, measured with
bench.py's own routine, alternating, in ONE process (boxes differ by several %)."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


class Args:   # the fields bench.measure_env_steps reads
    envs, mesh, rtol, env_groups, s1_steps, s1_warmup, no_flow_overlap, host_step, s1_solver_steps = 128, "ys930", 1e-10, 1, 50, 8, False, False, 500


dev = torch.device("cuda")
for rep in range(5):
    for fp in ("cg", "direct"):
        r = bench.measure_env_steps(Args, dev, None, 1, 1, steps=50, repeats=3, flow_pressure=fp)
        print(f"rep {rep} flow_pressure={fp:6s}: {r['ms_per_batched_step']:.3f} ms per batched step ({r['value']:.0f} env-steps/s), "
              f"pressure iterations {r['krylov_iters_per_ipcs_step']['pressure_cg']:.1f}", flush=True)
