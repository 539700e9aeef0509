"""Debug: per-phase s_memtime cycles of remesh_kernel<ACT> inside device-resident rollouts (library built with -DMDQ_RM_TRACE:
tools/micro/build_variant.sh rmtrace -DMDQ_RM_TRACE; MDQ_LIB_PATH=tools/micro/bin/libmdq_rmtrace.so python tools/trace_remesh.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd import _lib
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, os.environ.get("MDQ_TOOL_MESH", "ys930") + ".npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True, reproducible=False, rtol=1e-10)),
           agent_params=dict(solver_steps=int(os.environ.get("MDQ_TOOL_SOLVER_STEPS", "200")), episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=max(1, int(os.environ.get("MDQ_TOOL_SOLVER_STEPS", "200")) // 5), goal_vertices=0.95, plot_dir=""))
B = 128
venv = VecEnv2DAirfoil(cfg, B, flow_steps=0)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda(); fused = FusedGcn(net)
venv.get_state()
rng = np.random.default_rng(1370)
def run(k):
    ex = np.array([rng.random(B) < 0.5 for _ in range(k)]); ra = np.array([rng.integers(0, 181, B) for _ in range(k)])
    return venv.rollout_device(fused, k, ex, ra)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 16)()
run(10); torch.cuda.synchronize(); lib.mdq_rm_trace_host(buf, 1)
n = 30
run(n); torch.cuda.synchronize(); lib.mdq_rm_trace_host(buf, 0)
names = ["action head (Q-row argmax, decoding)", "stage coordinates, orient cells, star", "lane 0: ring, ear clipping, slots", "renumber + move coordinates",
         "neighbour table (hash)", "empty-circle test of every interior edge", "lane 0: Lawson flips", "canonical cells + write-back"]
tot = sum(buf[:8])
print(f"remesh_kernel<ACT>, mesh 0: {tot / n:.0f} ticks per launch")
for k, nm in enumerate(names):
    print(f"{k} {nm:45s} {buf[k] / n:9.0f}  {100.0 * buf[k] / max(tot, 1):5.1f} %")
