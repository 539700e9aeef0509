"""S3 / S1 env step with the main chain and the flow leg on disjoint halves of the chip (MDQ_CU_PARTITION) against the
unrestricted streams: run once per setting, e.g.
    MDQ_CU_PARTITION=0 python tools/try_cumask.py; MDQ_CU_PARTITION=1 python tools/try_cumask.py"""
import os, sys, time
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(1)
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnvGroups
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
from meshdqn_amd import streams
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
base = Env2DAirfoil(cfg)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda()
_pcg = os.environ.get("TRY_PCG_DEGREE")
variants = [("S1", dict(flow_steps=0)), ("S3", dict(flow_steps=1, flow_overlap=True, **(dict(flow_pcg_degree=int(_pcg)) if _pcg else {})))]
rng = np.random.default_rng(1370)
for name, kw in variants:
    grp = VecEnvGroups(cfg, B, 1, base_env=base, **kw)
    fused = FusedGcn(net)
    def run(k):
        ex = [np.array([rng.random(B) < 0.5 for _ in range(k)])]
        ra = [np.array([rng.integers(0, 181, B) for _ in range(k)])]
        grp.rollout_device([fused], k, ex, ra)
    run(10)
    rates = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(K)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rates.append(B * K / dt)
    print(f"MDQ_CU_PARTITION={os.environ.get('MDQ_CU_PARTITION', '0')} {name}: " + " ".join(f"{r:.0f}" for r in rates) + f"  best {max(rates):.0f} env-steps/s ({B / max(rates) * 1e3:.3f} ms)", flush=True)
print(streams.LOG)
