import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
m = np.load(os.path.join(R, "tests/golden/ys930.npz"))
t = MeshTopology(m["coords"], m["cells"]); x = smooth_coords(t, 50)
for mode, rtol in ((-2, 1e-13), (3, 1e-13), (3, 1e-10)):
    B = 45
    b = IpcsBatch([t] * B, [x] * B, rtol=rtol, mode=mode)
    b.evolve(100); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): d, l = b.evolve(100)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it = b.iters.cpu().numpy()[0] / 1100.0
    print(f"mode {mode} rtol {rtol:g}: {dt/1000*1e3:.4f} ms per step of {B} meshes; iterations per step {it}; drag {d[0,-1].item():.10f}")
