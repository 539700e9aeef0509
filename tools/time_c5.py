"""S2 rate on BASELINE config 5 (ys930 red-refined once: 6280 triangles, 25 848 velocity dofs) - assembled SELL path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.mesh_ops import red_refine
from meshdqn_amd.topology import MeshTopology
PMC = len(sys.argv) > 1 and sys.argv[1] == "pmc"     # counter passes (tools/refresh_profiles_r03.sh): short, one variant
B = int(sys.argv[1]) if len(sys.argv) > 1 and not PMC else 128
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ys930.npz"))
t0 = MeshTopology(z["coords"], z["cells"])
rc, rcells = red_refine(smooth_coords(t0, 50), z["cells"])
topo = MeshTopology(rc, rcells)
print("refined mesh", topo.nv, topo.nt, topo.ne, flush=True)
for direct in ((True,) if PMC else (True, False)):
    t = time.time()
    batch = IpcsBatch([topo] * B, [rc] * B, rtol=1e-10, pressure_direct=direct, mode=int(os.environ.get("MDQ_MODE", "-1")),
                      cell_order=os.environ.get("MDQ_CELL_ORDER", "auto"), pcg_degree=int(os.environ.get("MDQ_PCG", "0")))
    print("cell order", os.environ.get("MDQ_CELL_ORDER", "auto"), "NRL", batch.desc.NRL, "lpos", bool(batch.desc.mf_lpos), flush=True)
    batch.assemble(); torch.cuda.synchronize()
    print("setup s", round(time.time() - t, 1), "mode", batch.desc.mode, flush=True)
    out = (torch.empty((B, 1), dtype=torch.float64, device="cuda"), torch.empty((B, 1), dtype=torch.float64, device="cuda"))
    for _ in range(int(os.environ.get("SPINUP", "30" if PMC else "300"))):
        batch.evolve(1, out=out)
    batch.iters.zero_(); torch.cuda.synchronize(); n = 5 if PMC else 20; t = time.time()
    for _ in range(n):
        batch.evolve(1, out=out)
    torch.cuda.synchronize(); dt = time.time() - t
    it = batch.iters.cpu().numpy().astype(float) / n
    by = batch.algorithmic_bytes_per_step(it)
    print(f"B={B} direct={direct}: {dt/n*1e3:.2f} ms/step -> {B*n/dt:.0f} env-steps/s; iters {it.mean(0)}; "
          f"CSR-convention bytes/step {by/1e9:.2f} GB -> {by/(dt/n)/1e12:.2f} TB/s; drag {out[0][0,0].item():.6f}", flush=True)
    del batch
