import sys, time, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from meshdqn_amd.topology import MeshTopology
from meshdqn_amd.ipcs_batch import smooth_coords
from meshdqn_amd.mesh_ops import remesh_batch
z=np.load('tests/golden/ys930.npz'); t=MeshTopology(z['coords'],z['cells']); x=smooth_coords(t,50)
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
try: print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print('no cpu.max', e)
B=128
rng=np.random.default_rng(0)
for T in (1,8,32,64,128):
    best=1e9
    for rep in range(3):
        cb=np.tile(x[None],(B,1,1)).copy(); tb=np.tile(t.cells.astype(np.int32)[None],(B,1,1)).copy()
        nv=np.full(B,t.nv,np.int32); nt=np.full(B,t.nt,np.int32)
        rem=rng.integers(200,800,B).astype(np.int32)
        t0=time.time(); st=remesh_batch(cb,tb,nv,nt,rem,50,T); best=min(best,time.time()-t0)
    print('threads',T,'ms per call',best*1e3)
