"""Debug: per-section s_memtime cycles of topology_kernel (library built with MDQ_CFLAGS=-DMDQ_TOPO_TRACE).
usage: python tools/trace_topo.py [flow_steps]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshdqn_amd import _lib
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
FLOW = int(sys.argv[1]) if len(sys.argv) > 1 else 0
G = os.path.join(ROOT, "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, os.environ.get("MDQ_TOOL_MESH", "ys930") + ".npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=int(os.environ.get("MDQ_TOOL_SOLVER_STEPS", "200")), episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=max(1, int(os.environ.get("MDQ_TOOL_SOLVER_STEPS", "200")) // 5), goal_vertices=0.95, plot_dir=""))
venv = VecEnv2DAirfoil(cfg, 128, flow_steps=FLOW)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 16)()
rng = np.random.default_rng(0)
for _ in range(3):
    venv.step(rng.integers(0, 181, 128))
torch.cuda.synchronize(); lib.mdq_topo_trace_host(buf, 1)
n = 10
for _ in range(n):
    venv.step(rng.integers(0, 181, 128))
torch.cuda.synchronize(); lib.mdq_topo_trace_host(buf, 0)
names = ["load + hash init", "edges (hash, first-appearance ids)", "P2 points, boundary, facet tags", "removable", "polygon distances + argsort + window",
         "state graph", "IPCS: SELL pattern (tail) + [distances: segments, groups, chords]", "IPCS: Dirichlet data + [distances: home groups, keys]",
         "IPCS: outflow entries + [distances: counting sort]", "IPCS: packed metadata",
         "IPCS: gather lists P1", "IPCS: gather lists P2", "(of 4) polygon distances: candidates, exact evaluations, inside test (+ slots 6-8 in brackets)", "(of 4) argsort (bucket sort)", "(of 8) facet scan + entries", "(of 8) entry sort"]
tot = sum(buf[:16])
print(f"topology_kernel, mesh 0: {tot / n:.0f} ticks per launch")
for k, nm in enumerate(names):
    print(f"{k:2d} {nm:48s} {buf[k] / n:9.0f}  {100.0 * buf[k] / max(tot, 1):5.1f} %")
