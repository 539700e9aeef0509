"""The S3 step with the Jacobi-CG pressure solve vs the per-step device re-factorisation + direct solve, measured with
bench.py's own routine, alternating, in ONE process (boxes differ by several %)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
def get_args():
    # bench.main() parses inside main; replicate minimal fields used by measure_env_steps
    class A: pass
    a = A()
    a.envs, a.mesh, a.rtol, a.env_groups, a.s1_steps, a.s1_warmup, a.no_flow_overlap, a.host_step = 128, "ys930", 1e-10, 1, 50, 8, False, False
    a.s1_solver_steps = 500
    return a
args = get_args()
dev = torch.device("cuda")
for rep in range(5):
    for fp in ("cg", "direct"):
        r = bench.measure_env_steps(args, dev, None, 1, 1, steps=50, repeats=3, flow_pressure=fp)
        print(f"rep {rep} flow_pressure={fp:6s}: {r['ms_per_batched_step']:.3f} ms per batched step ({r['value']:.0f} env-steps/s), "
              f"pressure iterations {r['krylov_iters_per_ipcs_step']['pressure_cg']:.1f}", flush=True)
