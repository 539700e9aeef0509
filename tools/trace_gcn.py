"""Debug: per-level / per-phase s_memtime cycles of gcn_embed_kernel (library built with MDQ_CFLAGS=-DMDQ_GCN_TRACE)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshdqn_amd import _lib
from meshdqn_amd.airfoilgcnn import NodeRemovalNet
from meshdqn_amd.gcn_fused import FusedGcn
B, N, EM = 128, 180, 1536
rng = np.random.default_rng(0)
net = NodeRemovalNet(181, conv_width=128, topk=0.1); net.set_num_nodes(17); net = net.cuda(); fused = FusedGcn(net)
cnt = rng.integers(350, 500, size=B); ep = np.zeros(B + 1, np.int32); ep[1:] = np.cumsum(cnt)
x = torch.from_numpy(rng.standard_normal((B, N, 17))).float().cuda()
esrc = torch.from_numpy(rng.integers(0, N, size=int(ep[-1]))).int().cuda()
edst = torch.from_numpy(rng.integers(0, N, size=int(ep[-1]))).int().cuda()
node_ptr = torch.arange(B + 1, dtype=torch.int32, device="cuda") * N
eptr = torch.from_numpy(ep).cuda()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 80)()
for _ in range(3):
    fused.forward_arrays(x, node_ptr, esrc, edst, eptr, N, EM)
torch.cuda.synchronize(); lib.mdq_gcn_trace_host(buf, 1)
n = 20
for _ in range(n):
    fused.forward_arrays(x, node_ptr, esrc, edst, eptr, N, EM)
torch.cuda.synchronize(); lib.mdq_gcn_trace_host(buf, 0)
names = ["build_csr", "aggregation", "dense (conv)", "relu + score", "top-k rank", "pooled features", "edge filter", "readout", "(norm)"]
tot = sum(buf[:80])
print(f"total {tot / n:.0f} ticks per launch (graph 0)")
for l in range(4):
    print("level", l, " ".join(f"{names[k]}={buf[l * 10 + k] / n:.0f}" for k in range(9)))
