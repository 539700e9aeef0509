"""Launch time of the large-mesh smoothing (mdq_smooth_fast on the red-refined ys930, 50 sweeps) for B meshes, HIP events:
python tools/time_smooth_big.py [B]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, torch
from meshdqn_amd.ipcs_batch import smooth_coords
from meshdqn_amd.mesh_ops import red_refine, smooth_batch_gpu
from meshdqn_amd.topology import MeshTopology
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
m = np.load(os.path.join(R, "tests/golden/ys930.npz"))
rc, rcells = red_refine(smooth_coords(MeshTopology(m["coords"], m["cells"]), 50), m["cells"])
t = MeshTopology(rc, rcells)
host = smooth_coords(t, 50)
tt = torch.from_numpy(np.sort(rcells, axis=1).astype(np.int32)[None].repeat(B, 0).copy()).cuda()
one = lambda v: torch.full((B,), v, dtype=torch.int32, device="cuda")
for nsw in (50, 1):
  its = one(nsw)
  for fast in (True,):
    ms = []
    for rep in range(6):
        tc = torch.from_numpy(np.repeat(rc[None], B, 0).copy()).cuda()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); smooth_batch_gpu(tc, tt, one(t.nv), one(t.nt), its, fast=fast); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    err = np.abs(tc.cpu().numpy() - (host if nsw == 50 else smooth_coords(t, nsw))[None]).max()
    print(f"B={B} sweeps={nsw} fast={fast}: {np.median(ms[1:]):.3f} ms per launch (first {ms[0]:.3f}); max |x - host loop| {err:.2e}")
