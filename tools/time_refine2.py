"""ys930 red-refined once / twice: IPCS step time and drag of ONE env (dev tool for the resolution sweep)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.mesh_ops import red_refine
from meshdqn_amd.topology import MeshTopology
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
z = np.load(os.path.join(G, "ys930.npz"))
c, t = smooth_coords(MeshTopology(z["coords"], z["cells"]), 50), z["cells"]
for level in (1, 2):
    c, t = red_refine(c, t)
    t0 = time.time(); topo = MeshTopology(c, t); b = IpcsBatch([topo], [c], rtol=1e-10); b.assemble(); torch.cuda.synchronize()
    print(f"level {level}: {topo.nv} vertices {topo.nt} triangles, mode {int(b.desc.mode)}, setup {time.time() - t0:.1f} s", flush=True)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    t0 = time.time(); d, l = b.evolve(n); torch.cuda.synchronize(); dt = time.time() - t0
    it = b.iters.cpu().numpy()[0] / n
    print(f"  {dt / n * 1e3:.2f} ms per step, iterations {it}, drag {d[0, -1].item():.6f} lift {l[0, -1].item():.6f}", flush=True)
