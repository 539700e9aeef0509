"""Quick timing of the batched IPCS kernel (dev tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
z = np.load("tests/golden/ys930.npz")
t = MeshTopology(z["coords"], z["cells"])
x = smooth_coords(t, 50)
t0 = time.time()
batch = IpcsBatch([t] * B, [x] * B)
print("setup", time.time() - t0)
batch.assemble(); torch.cuda.synchronize()
t0 = time.time(); batch.assemble(); torch.cuda.synchronize(); print("assemble ms", (time.time() - t0) * 1e3)
d, l = batch.evolve(5); torch.cuda.synchronize()
print("drag after 5", d[0].tolist())
batch.iters.zero_()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    d, l = batch.evolve(nsteps); torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"B={B} nsteps={nsteps}: {dt*1e3/nsteps:.3f} ms/step  {B*nsteps/dt:.0f} env-steps/s")
it = batch.iters.cpu().numpy().astype(float) / (3 * nsteps)
print("iters/step (u,p,m): mean", it.mean(0), "max", it.max(0))
