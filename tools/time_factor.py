"""Duration of mdq_ipcs_factorize_pressure (device-side substructuring factors) for B ys930 meshes, and the IPCS step
with the direct pressure solve on those factors next to the Jacobi-CG one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.topology import MeshTopology
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ys930.npz"))
topo = MeshTopology(z["coords"], z["cells"])
x = smooth_coords(topo, 50)
for pd in ("device", True, False):
    batch = IpcsBatch([topo] * B, [x] * B, device="cuda", pressure_direct=pd)
    batch.evolve(50)
    torch.cuda.synchronize()
    if pd == "device":
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        batch.factorize_pressure_device()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            batch.factorize_pressure_device()
        e1.record(); torch.cuda.synchronize()
        import ctypes
        from meshdqn_amd import _lib
        L = ctypes.CDLL(_lib.LIB_PATH)
        if hasattr(L, "mdq_pf_trace_host"):
            buf = (ctypes.c_longlong * 16)()
            L.mdq_pf_trace_host(buf, 0)
            names = ["bisection", "separator + ordering", "K[G,I] CSR + S init", "subdomain: K_II / K_IG fill", "subdomain: Gauss-Jordan",
                     "subdomain: W out + F", "subdomain: Schur update", "S inverse"]
            tot = sum(buf[:8])
            print("phases of mesh 0 (share of the kernel):", ", ".join(f"{n} {100.0 * buf[i] / max(tot, 1):.1f} %" for i, n in enumerate(names)))
        print(f"mdq_ipcs_factorize_pressure, {B} ys930 meshes: {e0.elapsed_time(e1) / 20:.3f} ms per launch; status {batch.pd_status.unique().tolist()}, "
              f"header (nI, nG, parts) {batch.t['pd_hdr'][0].tolist()}")
    t0 = time.perf_counter()
    for _ in range(200):
        batch.evolve(1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"pressure_direct={pd!s:7s}: {dt * 1e3:.3f} ms per IPCS step of {B} envs, pressure iterations per step {batch.iters[:, 1].float().mean().item() / 250:.1f}")
