"""Debug: s_memtime cycles of the sections of a level of smooth_big_kernel (library built with MDQ_CFLAGS=-DMDQ_SB_TRACE):
python tools/trace_smooth_big.py"""
import ctypes as C, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, torch
from meshdqn_amd import _lib
from meshdqn_amd.ipcs_batch import smooth_coords
from meshdqn_amd.mesh_ops import red_refine, smooth_batch_gpu
from meshdqn_amd.topology import MeshTopology
lib = _lib.load()
fn = lib.mdq_sb_trace_host
fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
B = 128
m = np.load(os.path.join(R, "tests/golden/ys930.npz"))
rc, rcells = red_refine(smooth_coords(MeshTopology(m["coords"], m["cells"]), 50), m["cells"])
t = MeshTopology(rc, rcells)
tt = torch.from_numpy(np.sort(rcells, axis=1).astype(np.int32)[None].repeat(B, 0).copy()).cuda()
one = lambda v: torch.full((B,), v, dtype=torch.int32, device="cuda")
out = (C.c_longlong * 8)()
for rep in range(2):
    tc = torch.from_numpy(np.repeat(rc[None], B, 0).copy()).cuda()
    fn(out, 1)
    smooth_batch_gpu(tc, tt, one(t.nv), one(t.nt), one(50)); torch.cuda.synchronize()
    fn(out, 0)
names = ["level starts + record request", "updates", "rotate (waits for the LDS level starts)", "barrier"]
tot = sum(out[k] for k in range(4))
print("cycles of wave 0, mesh 0, per launch (100 MHz s_memtime ticks x 24 = 2.4 GHz cycles):")
for k in range(4):
    print(f"  {names[k]:45s} {out[k]:8d} ticks = {out[k] * 10 / 2000:.1f} ns per level")
print(f"  total {tot} ticks = {tot / 100:.1f} us")
