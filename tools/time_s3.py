"""A/B of the S3 env step in ONE process (boxes differ by ~10 %): in-line IPCS step vs IPCS step on the flow stream."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = dict(flow_config=dict(flow_params=dict(mu=1e-3, rho=1.0, inflow="constant"), geometry_params=dict(mesh=os.path.join(G, "ys930.npz")),
                            solver_params=dict(dt=0.001, solver_type="lu", smooth=True)),
           agent_params=dict(solver_steps=500, episodes=10, timesteps=10000, threshold=0.001, N_closest=180, gt_drag=-1, gt_time=-1, u=-1, p=-1,
                             time_reward=0.005, save_steps=100, goal_vertices=0.95, plot_dir=""))
base = Env2DAirfoil(cfg)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda()
variants = [("S1", dict(flow_steps=0)), ("S3 inline", dict(flow_steps=1)), ("S3 overlap", dict(flow_steps=1, flow_overlap=True))]
groups = {n: VecEnvGroups(cfg, B, 1, base_env=base, **kw) for n, kw in variants}
fused = {n: FusedGcn(net) for n in groups}
rngs = {n: np.random.default_rng(1370) for n in groups}
def act_for(n):
    def act(g, env, st):
        q = fused[n].forward_arrays(st["x"], st["node_ptr"], st["esrc"], st["edst"], st["edge_ptr"], env.N, env.EMAX)
        greedy = q.argmax(1).cpu().numpy()
        return np.where(rngs[n].random(env.B) < 0.5, rngs[n].integers(0, 181, env.B), greedy)
    return act
for n in groups:
    groups[n].rollout(act_for(n), 30)
def dev_run(n, k):
    ex = [np.array([rngs[n].random(B) < 0.5 for _ in range(k)])]
    ra = [np.array([rngs[n].integers(0, 181, B) for _ in range(k)])]
    groups[n].rollout_device([fused[n]], k, ex, ra)
for n in groups:
    dev_run(n, 10)
for rep in range(4):
    for n in groups:
        for mode in ("host", "device"):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            groups[n].rollout(act_for(n), K) if mode == "host" else dev_run(n, K)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"rep {rep} {n:12s} {mode:6s}: {dt / K * 1e3:.3f} ms per batched step -> {B * K / dt:.0f} env-steps/s", flush=True)
# in-rollout duration of the smoothing launches (HIP events) per variant
for n in groups:
    env = groups[n].envs[0]
    env.smooth_events = []
    from meshdqn_amd import _lib as _L
    _L.load().mdq_smooth_stats(None, 1)
    dev_run(n, 30)
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in env.smooth_events])
    env.smooth_events = None
    print(f"{n:12s}: smooth_kernel in the rollout {ms.mean():.3f} ms (min {ms.min():.3f}, max {ms.max():.3f})")
    st = np.zeros(64, np.int64)
    _L.load().mdq_smooth_stats(st.ctypes.data, 0)
    print(f"{'':12s}  {len(ms)} launches; abandoned speculative sweeps by sweep index: {dict((int(i), int(c)) for i, c in enumerate(st) if c)}")
    if st[46]:
        f32 = lambda u: float(np.array([u], np.uint32).view(np.float32)[0])
        print(f"{'':12s}  last undecided update: env {st[40]} vertex {st[41]} of {st[46]} sweep {st[45]} q2 {f32(st[42]):.4e} "
              f"r_min^2 {f32(st[43]):.4e} lower limit {f32(st[44]):.4e}")
    print(f"{'':12s}  launch durations (ms) deciles: {np.round(np.percentile(ms, np.arange(0, 101, 10)), 3).tolist()}")
