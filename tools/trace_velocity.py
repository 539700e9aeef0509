"""Debug: per-phase s_memtime cycles of at_velocity_kernel (library built with MDQ_CFLAGS=-DMDQ_AT_TRACE).
usage: python tools/trace_velocity.py [envs] [steps]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshdqn_amd import _lib
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
z = np.load(os.path.join(ROOT, "tests", "golden", "ys930.npz"))
topo = MeshTopology(z["coords"], z["cells"])
x = smooth_coords(topo, 50)
batch = IpcsBatch([topo] * B, [x] * B, device="cuda")
batch.assemble()
for _ in range(300):
    batch.evolve(1)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 16)()
lib.mdq_at_trace_host(buf, 1)
lib.mdq_ct_trace_host((ctypes.c_longlong * 16)(), 1)
batch.iters.zero_()
for _ in range(steps):
    batch.evolve(1)
torch.cuda.synchronize()
lib.mdq_at_trace_host(buf, 0)
it = batch.iters.cpu().numpy()[:, 0].mean() / steps
_, _, kms = batch.evolve_timed(steps)
print("HIP-event kernel ms/step (velocity, pressure, correction):", [round(m / steps, 4) for m in kms])
names = ["prologue (prefetch, outflow rows)", "zero + rhs1 element loop", "outflow + x0 extrapolation/history", "A x0 element loop",
         "r0, 2 reductions, p=0", "it: p update + barrier", "it: A p element loop", "it: v, (rh,v) reduction",
         "it: s update + barrier", "it: A s element loop (+xs loads)", "it: t, 2 dots reduction", "it: x update, 2 dots reduction",
         "tail barrier"]
tot = sum(buf[:13])
print(f"B={B} steps={steps} bicgstab iters/step={it:.2f}  total {tot / steps:.0f} ticks/step")
for k, n in enumerate(names):
    print(f"{k:2d} {n:40s} {buf[k] / steps:9.0f} ticks/step  {100.0 * buf[k] / tot:5.1f} %")

cb = (ctypes.c_longlong * 16)()
lib.mdq_ct_trace_host(cb, 0)
cn = ["prologue", "zero + rhs3 element loop", "f3 / x0 / lift phase", "M x0 element loop", "r, p init + 2 reductions",
      "it: stage p + barrier", "it: M p element loop", "it: q, (p,q) reduction", "it: x, r update + reduction",
      "state update + barrier", "forces"]
tot = sum(cb[:11])
n_all = steps + steps     # (evolve + evolve_timed passes)
print(f"correction kernel: total {tot / n_all:.0f} ticks/step")
for k, n in enumerate(cn):
    print(f"{k:2d} {n:40s} {cb[k] / n_all:9.0f} ticks/step  {100.0 * cb[k] / tot:5.1f} %")

pb = (ctypes.c_longlong * 16)()
lib.mdq_pt_trace_host(pb, 0)
pn = ["rhs2 element loop + row gather (before the solve)", "permute b", "y_I = W b_I", "g = b_G - K_GI y_I", "x_G = Sinv g (+ slice sum)", "x_I = y_I - F x_G, scatter"]
n_all = 300 + 3 * steps
tot = sum(pb[:6])
print(f"pressure kernel (direct): total {tot / n_all:.0f} ticks/step (all launches since process start: {n_all})")
for k, n in enumerate(pn):
    print(f"{k:2d} {n:52s} {pb[k] / n_all:9.0f} ticks/step  {100.0 * pb[k] / max(tot, 1):5.1f} %")
