"""Duration of mdq_remesh for 128 ys930 meshes (one random interior vertex removed each), HIP events; compare library
variants through MDQ_LIB_PATH (tools/micro/build_variant.sh NAME -DMDQ_REMESH_WG=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.mesh_ops import remesh_batch_gpu
from meshdqn_amd.topology import MeshTopology
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(os.path.join(ROOT, "tests", "golden", "ys930.npz"))
B = 128
topo = MeshTopology(z["coords"], z["cells"])
interior = np.flatnonzero(~topo.on_boundary_vertices()) if hasattr(topo, "on_boundary_vertices") else np.arange(300, 800)
rng = np.random.default_rng(0)
c0 = torch.from_numpy(np.repeat(z["coords"][None], B, 0).copy()).cuda()
t0 = torch.from_numpy(np.repeat(z["cells"][None].astype(np.int32), B, 0).copy()).cuda()
ms = []
for rep in range(12):
    coords, cells = c0.clone(), t0.clone()
    nv = torch.full((B,), z["coords"].shape[0], dtype=torch.int32, device="cuda")
    nt = torch.full((B,), z["cells"].shape[0], dtype=torch.int32, device="cuda")
    rem = torch.from_numpy(rng.choice(interior, B).astype(np.int32)).cuda()
    st = torch.zeros(B, dtype=torch.int32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    remesh_batch_gpu(coords, cells, nv, nt, rem, st)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
    ok = int((st == 0).sum())
print(f"{os.environ.get('MDQ_LIB_PATH', 'default')}: remesh of {B} meshes {np.median(ms[2:]) * 1e3:.1f} us (min {min(ms[2:]) * 1e3:.1f}); {ok} of {B} removals succeeded in the last launch")
