import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.mesh_ops import smooth_batch_gpu
from meshdqn_amd.topology import MeshTopology
B = 128
z = np.load("/root/repo/tests/golden/ys930.npz")
topo = MeshTopology(z["coords"], z["cells"])
x = smooth_coords(topo, 50)
batch = IpcsBatch([topo] * B, [x] * B, device="cuda", pressure_direct="device")
batch.evolve(5)
dev = torch.device("cuda")
tc = torch.from_numpy(np.repeat(z["coords"][None], B, 0).copy()).cuda()
cells = torch.from_numpy(np.repeat(np.sort(z["cells"], axis=1)[None].astype(np.int32), B, 0).copy()).cuda()
nv = torch.full((B,), topo.nv, dtype=torch.int32, device=dev); nt = torch.full((B,), topo.nt, dtype=torch.int32, device=dev)
its = torch.full((B,), 50, dtype=torch.int32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def ev(): return torch.cuda.Event(enable_timing=True)
for mode in ("factor alone", "factor beside smoothing", "velocity-ish evolve beside smoothing", "evolve alone"):
    res = []
    for rep in range(5):
        torch.cuda.synchronize()
        a0, a1, b0, b1 = ev(), ev(), ev(), ev()
        if "beside" in mode:
            with torch.cuda.stream(s1):
                a0.record(); smooth_batch_gpu(tc.clone(), cells, nv, nt, its); a1.record()
        with torch.cuda.stream(s2):
            time.sleep(0.0002)
            b0.record()
            if "factor" in mode: batch.factorize_pressure_device()
            else: batch.evolve(1)
            b1.record()
        torch.cuda.synchronize()
        res.append((b0.elapsed_time(b1), a0.elapsed_time(a1) if "beside" in mode else 0.0))
    print(f"{mode:40s}: {np.median([r[0] for r in res]):.3f} ms (smoothing {np.median([r[1] for r in res]):.3f} ms)")
