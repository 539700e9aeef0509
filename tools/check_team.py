"""mode 4 (two workgroups per environment) against mode 0 (one) on the refined mesh, and both against the golden first
steps on ys930 (dev tool; the pytest versions live in tests/test_ipcs_gpu.py)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.mesh_ops import red_refine
from meshdqn_amd.topology import MeshTopology
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z = np.load(os.path.join(G, "ys930.npz"))
t0 = MeshTopology(z["coords"], z["cells"])
x0 = smooth_coords(t0, 50)
flow = json.load(open(os.path.join(G, "oracle_flow.json")))["ys930"]["steps"]
for mode in (0, 4):
    b = IpcsBatch([t0] * 3, [x0] * 3, rtol=1e-12, mode=mode)
    for s in (1, 2, 3):
        d, l = b.evolve(1)
        torch.cuda.synchronize()
        g = flow[str(s)]
        print(f"ys930 mode {mode} step {s}: drag rel err {abs(d[0, 0].item() - g['drag']) / abs(g['drag']):.2e} lift {abs(l[2, 0].item() - g['lift']) / abs(g['lift']):.2e}", flush=True)
rc, rcells = red_refine(x0, z["cells"])
rt = MeshTopology(rc, rcells)
res = {}
for mode in (0, 4):
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    b = IpcsBatch([rt] * B, [rc] * B, rtol=1e-10, mode=mode)
    b.assemble(); torch.cuda.synchronize()
    t = time.time(); d, l = b.evolve(10); torch.cuda.synchronize(); dt = time.time() - t
    res[mode] = (d.cpu().numpy(), l.cpu().numpy(), b.u_n.cpu().numpy(), b.iters.cpu().numpy())
    print(f"refined B={B} mode {mode}: {dt / 10 * 1e3:.2f} ms/step drag {d[0, -1].item():.10f} iters {b.iters.cpu().numpy()[0] / 10}", flush=True)
print("mode 4 vs 0: drag", np.abs(res[4][0] - res[0][0]).max() / np.abs(res[0][0]).max(), "u", np.abs(res[4][2] - res[0][2]).max() / np.abs(res[0][2]).max(),
      "all envs equal:", bool((res[4][0] == res[4][0][0]).all()))
