"""Debug: where the cycles of one block step of smooth_linear_kernel go (library built with -DMDQ_LIN_TRACE2:
tools/micro/build_variant.sh lintrace2 -DMDQ_LIN_TRACE2; MDQ_LIB_PATH=tools/micro/bin/libmdq_lintrace2.so python tools/trace_smooth_block.py)."""
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
buf = (ctypes.c_longlong * 8)()
for _ in range(2):
    smooth_batch_gpu(coords.clone(), cells, nv, nt, it)
torch.cuda.synchronize(); lib.mdq_lin_bt_host(buf, 1)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cs = [coords.clone() for _ in range(n)]
e0.record()
for c in cs:
    smooth_batch_gpu(c, cells, nv, nt, it)
e1.record()
torch.cuda.synchronize(); lib.mdq_lin_bt_host(buf, 0)
names = ["gather: 7 LDS reads + 6 adds", "cross-half add (permlane32 swap) + store of g", "8 x ds_read_b128 of g", "2 chains of 8 FMAs (+ wait for the M rows)",
         "cross-half add + store of x"]
nsteps = n * 50 * 22           # block steps of the x wave of mesh 0 (ys930: 22 blocks)
tot = sum(buf[:5])
print(f"kernel {e0.elapsed_time(e1) / n * 1e3:.0f} us per launch (trace build: every stamp is a fence); block step: {tot / nsteps:.0f} s_memtime cycles")
for k, nm in enumerate(names):
    print(f"{k} {nm:52s} {buf[k] / nsteps:7.1f} cycles  {100.0 * buf[k] / max(tot, 1):5.1f} %")
