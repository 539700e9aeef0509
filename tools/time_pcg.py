"""Pressure CG of the three-kernel mode on the ys930 mesh (B envs, developed flow, no direct solve): kernel time and
iterations for Chebyshev degrees of the polynomial preconditioner (0 = the plain Jacobi-CG kernel).  Dev tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
z = np.load(os.path.join(G, "ys930.npz"))
topo = MeshTopology(z["coords"], z["cells"])
x = smooth_coords(topo, 50)
ref = None
for deg in (0, -1, 4, 8):
    b = IpcsBatch([topo] * B, [x] * B, rtol=1e-10, pressure_direct=False, pcg_degree=deg)
    b.assemble()
    for _ in range(60):
        d, l = b.evolve(1)
    b.iters.zero_()
    n = 40
    d, l, ms = b.evolve_timed(n)
    it = b.iters.cpu().numpy().astype(float)[0] / n
    dd = d[0, -1].item()
    ref = dd if ref is None else ref
    print(f"degree {deg}: pressure kernel {ms[1] / n * 1e3:7.1f} us  iterations {it[1]:6.1f}  (velocity {ms[0] / n * 1e3:.1f} us, correction "
          f"{ms[2] / n * 1e3:.1f} us)  drag {dd:.12f} (rel. diff to degree 0: {abs(dd - ref) / abs(ref):.1e})", flush=True)
