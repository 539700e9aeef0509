import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from meshdqn_amd.mesh_ops import smooth_batch_gpu
from meshdqn_amd.ipcs_batch import smooth_coords
from meshdqn_amd.topology import MeshTopology
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for name in ("ys930", "ah93w145"):
    z = np.load(os.path.join(R, "tests", "golden", name + ".npz"))
    t = MeshTopology(z["coords"], z["cells"])
    ref = smooth_coords(t, 50)
    B = 256
    c0 = torch.from_numpy(np.repeat(z["coords"][None], B, 0).copy()).cuda()
    cells = torch.from_numpy(np.repeat(np.sort(z["cells"], 1)[None].astype(np.int32), B, 0).copy()).cuda()
    nv = torch.full((B,), z["coords"].shape[0], dtype=torch.int32, device="cuda")
    nt = torch.full((B,), z["cells"].shape[0], dtype=torch.int32, device="cuda")
    it = torch.full((B,), 50, dtype=torch.int32, device="cuda")
    worst = 0.0
    for rep in range(20):
        c = c0.clone(); smooth_batch_gpu(c, cells, nv, nt, it); torch.cuda.synchronize()
        out = c.cpu().numpy()
        assert (out == out[0]).all(), "environments of one launch differ"
        if rep == 0: first = out[0].copy()
        assert (out[0] == first).all(), "launches differ"
        worst = max(worst, np.abs(out[0] - ref).max())
    print(name, "20 launches x 256 meshes bitwise identical; max |gpu - host| =", worst)
