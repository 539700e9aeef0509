"""Time mdq_smooth (GPU dataflow smoothing) for B copies of ys930 (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.mesh_ops import smooth_batch_gpu
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ys930.npz"))
for B in (1, 128, 256):
    c = torch.from_numpy(np.repeat(z["coords"][None], B, 0).copy()).cuda()
    t = torch.from_numpy(np.repeat(np.sort(z["cells"], 1)[None].astype(np.int32), B, 0).copy()).cuda()
    nv = torch.full((B,), z["coords"].shape[0], dtype=torch.int32, device="cuda")
    nt = torch.full((B,), z["cells"].shape[0], dtype=torch.int32, device="cuda")
    it = torch.full((B,), 50, dtype=torch.int32, device="cuda")
    for rep in range(3):
        cc = c.clone(); torch.cuda.synchronize(); t0 = time.time()
        smooth_batch_gpu(cc, t, nv, nt, it); torch.cuda.synchronize()
        print(f"B={B}: {1e3 * (time.time() - t0):.3f} ms")
