"""Break down one DQN optimiser step (select = True): eager vs HIP-graph replay (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from meshdqn_amd.data import Data
from meshdqn_amd.trainer import DistContext, DQNTrainer, Transition
rng = np.random.default_rng(4)
def graph(e):
    return Data(x=torch.from_numpy(rng.standard_normal((180, 17))).float().cuda(), edge_index=torch.from_numpy(rng.integers(0, 180, size=(2, e))).long().cuda())
trs = [Transition(graph(int(rng.integers(200, 600))), torch.tensor([[int(rng.integers(0, 181))]]), None if i % 5 == 0 else graph(int(rng.integers(200, 600))), torch.tensor([float(rng.standard_normal())])) for i in range(32)]
import cProfile, pstats
for use_graph in (False, True):
    tr = DQNTrainer(n_actions=180, num_inputs=17, ctx=DistContext())
    tr.graphs = use_graph; tr.select = True
    for _ in range(3):
        tr.num_grads = 1; tr.optimize(trs)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20):
        tr.num_grads = 1; tr.optimize(trs)
    torch.cuda.synchronize(); print("graph" if use_graph else "eager", f"{(time.time()-t0)/20*1e3:.2f} ms per optimise", tr._graph_error)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        tr.num_grads = 1; tr.optimize(trs)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(10)
