"""Debug: s_memtime cycles of the phases of smooth_linear_kernel (library built with -DMDQ_LIN_TRACE:
tools/micro/build_variant.sh lintrace -DMDQ_LIN_TRACE; MDQ_LIB_PATH=tools/micro/bin/libmdq_lintrace.so python tools/trace_smooth.py)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshdqn_amd import _lib
from meshdqn_amd.mesh_ops import smooth_batch_gpu
z = np.load(os.path.join(ROOT, "tests", "golden", "ys930.npz"))
B = 128
coords = torch.from_numpy(np.repeat(z["coords"][None], B, 0).copy()).cuda()
cells = torch.from_numpy(np.repeat(z["cells"][None].astype(np.int32), B, 0).copy()).cuda()
nv = torch.full((B,), z["coords"].shape[0], dtype=torch.int32, device="cuda")
nt = torch.full((B,), z["cells"].shape[0], dtype=torch.int32, device="cuda")
it = torch.full((B,), 50, dtype=torch.int32, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 16)()
for _ in range(2):
    smooth_batch_gpu(coords.clone(), cells, nv, nt, it)
torch.cuda.synchronize(); lib.mdq_lin_trace_host(buf, 1)
n = 10
for _ in range(n):
    smooth_batch_gpu(coords.clone(), cells, nv, nt, it)
torch.cuda.synchronize(); lib.mdq_lin_trace_host(buf, 0)
names = ["vertex -> cells (atomics, scans, fill)", "per vertex: cell sort, neighbour lists, interior test", "interior ranks (scan)",
         "per rank: gather slots, lower lists, validation flags", "block inverses -> workspace", "positions into LDS", "50 sweeps"]
tot = sum(buf[:7])
print(f"smooth_linear_kernel, mesh 0: {tot / n:.0f} ticks per launch")
for k, nm in enumerate(names):
    print(f"{k} {nm:58s} {buf[k] / n:9.0f}  {100.0 * buf[k] / max(tot, 1):5.1f} %")
