"""5000-step trajectories from rest on both lab meshes: operator mode x solver tolerance -> deviation from the golden
checkpoints (oracle, exact LU), bitwise repeatability of two runs, equality of a batch of two with two batches of one,
wall time.  usage: python tools/traj_determinism.py [modes] [rtols]   e.g. 2,3 1e-10,1e-13"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology

G = os.path.join(ROOT, "tests", "golden")
FLOW = json.load(open(os.path.join(G, "oracle_flow.json")))
names = ["ys930", "ah93w145"]
modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else "2,3").split(",")]
rtols = [float(r) for r in (sys.argv[2] if len(sys.argv) > 2 else "1e-10,1e-13").split(",")]
topos = [MeshTopology(*[np.load(os.path.join(G, f"{n}.npz"))[k] for k in ("coords", "cells")]) for n in names]
xs = [smooth_coords(t, 50) for t in topos]


def run(sel, mode, rtol, pd="device"):
    batch = IpcsBatch([topos[i] for i in sel], [xs[i] for i in sel], device="cuda", mode=mode, rtol=rtol, pressure_direct=pd)
    batch.assemble()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    D, L = [], []
    for k in range(5):
        d, l = batch.evolve(1000)
        D.append(d[:, -1].clone()); L.append(l[:, -1].clone())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return torch.stack(D, 1).cpu().numpy(), torch.stack(L, 1).cpu().numpy(), dt, batch.iters.cpu().numpy() / 5000.0


for mode in modes:
    for rtol in rtols:
        for pd in ("device", True):
            D1, L1, t1, it = run([0, 1], mode, rtol, pd)
            D2, L2, t2, _ = run([0, 1], mode, rtol, pd)
            Ds = np.concatenate([run([i], mode, rtol, pd)[0] for i in (0, 1)])
            dev = []
            for b, n in enumerate(names):
                for k in range(5):
                    g = FLOW[n]["steps"][str(1000 * (k + 1))]
                    dev.append(max(abs(D1[b, k] - g["drag"]) / abs(g["drag"]), abs(L1[b, k] - g["lift"]) / abs(g["lift"])))
            print(f"mode {mode} rtol {rtol:g} pressure {pd}: {t1:.2f} s / 5000 steps; vs golden " + " ".join(f"{v:.1e}" for v in dev)
                  + f"; run-to-run max rel {np.abs(D1 - D2).max() / np.abs(D1).max():.1e} bitwise {np.array_equal(D1, D2) and np.array_equal(L1, L2)}"
                  + f"; batch-of-2 vs singles max rel {np.abs(D1 - Ds).max() / np.abs(D1).max():.1e} bitwise {np.array_equal(D1, Ds)}"
                  + f"; iters/step {it.mean(0).round(2).tolist()}", flush=True)
