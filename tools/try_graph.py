"""Can a whole device-resident rollout (K env steps, main + flow streams) be captured as ONE HIP graph?  Dev experiment."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from meshdqn_amd.env import Env2DAirfoil
from meshdqn_amd.vec_env import VecEnv2DAirfoil
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 20
base = Env2DAirfoil(cfg)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda()
fg = FusedGcn(net); fg._pack()
rng = np.random.default_rng(3)
ex, ra = rng.random((K, B)) < 0.5, rng.integers(0, 181, (K, B))
env = VecEnv2DAirfoil(cfg, B, base_env=base, flow_steps=1, flow_overlap=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    env.calibrate_streams(fg)
    ref = env.rollout_device(fg, K, ex, ra)          # eager (also warms every lazily created buffer)
    env.flow_wait(); env.reset_all()
    torch.cuda.synchronize()
    ro = env.rollout_begin(K, ex, ra)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            for _ in range(K):
                env.rollout_step(ro, fg, pack=False)
            s.wait_stream(env._flow_stream)           # join the flow leg before the capture ends
        print("captured", flush=True)
    except Exception as exc:
        print("capture failed:", repr(exc)[:600], flush=True); sys.exit(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    out = env.rollout_end(ro)
    print(f"replay: host {1e3 * (t1 - t0) / K:.4f} ms per step, total {1e3 * (t2 - t0) / K:.3f} ms per step")
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"  replay {rep}: {1e3 * (t2 - t0) / K:.4f} ms per step -> {B * K / (t2 - t0):.0f} env-steps/s")
    for rep in range(3):
        ex2, ra2 = rng.random((K, B)) < 0.5, rng.integers(0, 181, (K, B))
        torch.cuda.synchronize(); t0 = time.perf_counter(); env.rollout_device(fg, K, ex2, ra2); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"  eager {rep}: {1e3 * (t2 - t0) / K:.4f} ms per step -> {B * K / (t2 - t0):.0f} env-steps/s")
    print("same actions", np.array_equal(out["actions"], ref["actions"]), "rewards", np.abs(out["rewards"] - ref["rewards"]).max(), "dones", np.array_equal(out["dones"], ref["dones"]))
