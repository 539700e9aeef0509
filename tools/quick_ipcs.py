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

if os.environ.get("MDQ_CFLAGS", "").find("MDQ_PROFILE") >= 0:
    c = batch.cap
    NV, NT, NE = c["NV"], c["NT"], c["NE"]
    N2 = NV + NE
    per = (12 * NT + 12 * N2 + NV + 24 + 31) // 32 * 32
    w = batch.t["work"].cpu().numpy().reshape(B, per)
    prof = w[:, 12 * NT + 12 * N2 + NV: 12 * NT + 12 * N2 + NV + 8]
    tot_steps = 5 + 3 * nsteps
    names = ["elem_rhs1", "gather1", "bicgstab", "rhs2+gather", "cg_pressure", "rhs3+gather", "cg_mass", "update+probe"]
    m = prof.mean(0) / tot_steps
    for n_, v_ in zip(names, m):
        print(f"  {n_:14s} {v_/100.0:9.1f} us/step (at 100MHz memtime)  {100*v_/m.sum():5.1f}%")
    tp = w[:, 12 * NT + 12 * N2 + NV + 8: 12 * NT + 12 * N2 + NV + 16].mean(0) / tot_steps
    print("  (mode3: elem-compute, atomics, prefetch) tile_accumulate phases (thread 0, all call sites): elem %.0f  prefetch-issue %.0f  barrier1 %.0f  gather %.0f  barrier2 %.0f  | total %.0f of step %.0f" % (tp[0], tp[1], tp[2], tp[3], tp[4], tp[:5].sum(), m.sum()))
    bp = w[:, 12 * NT + 12 * N2 + NV + 16: 12 * NT + 12 * N2 + NV + 24].mean(0) / tot_steps
    print("  bicgstab per step (cycles): [8] %.0f | [9] %.0f | [10] %.0f | [11] %.0f | [12] %.0f | [13] %.0f | [14] %.0f | [15] %.0f" % tuple(bp))
    print("   mode3 legend: 8 p-upd+sync, 9 apply1, 10 barrier after apply1, 11 readY+dot+red a1, 12 s-upd+sync, 13 apply2+xprefetch+sync+dots+red a3, 14 update+red a4, 15 loop top")
