"""Debug: cycles of the phases of a mode-5 operator application on the refined mesh (library built with -DMDQ_T5_TRACE:
tools/micro/build_variant.sh t5 -DMDQ_T5_TRACE; MDQ_LIB_PATH=tools/micro/bin/libmdq_t5.so python tools/trace_mode5.py [B])."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd import _lib
from meshdqn_amd.ipcs_batch import IpcsBatch, smooth_coords
from meshdqn_amd.mesh_ops import red_refine
from meshdqn_amd.topology import MeshTopology
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ys930.npz"))
t0 = MeshTopology(z["coords"], z["cells"])
rc, rcells = red_refine(smooth_coords(t0, 50), z["cells"])
topo = MeshTopology(rc, rcells)
batch = IpcsBatch([topo] * B, [rc] * B, rtol=1e-10, mode=int(os.environ.get("MDQ_MODE", "5")))
batch.assemble()
for _ in range(30):
    batch.evolve(1)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 8)()
torch.cuda.synchronize(); lib.mdq_t5_trace_host(buf, 1)
n = 10
t = time.time()
for _ in range(n):
    batch.evolve(1)
torch.cuda.synchronize(); dt = time.time() - t
lib.mdq_t5_trace_host(buf, 0)
napp = max(buf[7], 1)
names = ["element phase (metadata, gathers, operator, tile stores)", "barrier", "row phase (tile ranges, running sums)", "barrier", "epilogue pass over the rows"]
print(f"B={B}: {dt / n * 1e3:.2f} ms per step, {napp / n:.1f} applications per step, {sum(buf[:5]) / napp:.0f} cycles per application")
for k, nm in enumerate(names):
    print(f"{k} {nm:60s} {buf[k] / napp:9.0f} cycles per application  {100.0 * buf[k] / max(sum(buf[:5]), 1):5.1f} %")
