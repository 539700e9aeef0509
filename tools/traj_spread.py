"""Run-to-run spread of the 5000-step trajectory against the golden checkpoints (LDS atomics make mode 3 reproducible
to round-off only).  usage: python tools/traj_spread.py [repeats]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology

G = os.path.join(ROOT, "tests", "golden")
FLOW = json.load(open(os.path.join(G, "oracle_flow.json")))
names = ["ys930", "ah93w145"]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for r in range(reps):
    topos = [MeshTopology(*[np.load(os.path.join(G, f"{n}.npz"))[k] for k in ("coords", "cells")]) for n in names]
    xs = [smooth_coords(t, 50) for t in topos]
    batch = IpcsBatch(topos, xs, device="cuda")
    out = []
    for k in range(1, 6):
        for _ in range(10):
            drag, lift = batch.evolve(100)
        torch.cuda.synchronize()
        for b, n in enumerate(names):
            g = FLOW[n]["steps"][str(1000 * k)]
            out.append(max(abs(drag[b, -1].item() - g["drag"]) / abs(g["drag"]), abs(lift[b, -1].item() - g["lift"]) / abs(g["lift"])))
    print("run", r, "max rel dev per checkpoint (ys930, ah93w145 x 5):", " ".join(f"{v:.1e}" for v in out),
          "iters", (batch.iters.cpu().numpy().mean(0) / 5000).round(2).tolist(), flush=True)
