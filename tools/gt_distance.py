"""Distance of the product's 5000-step ground truth (reproducible mode, rtol 1e-13) from the oracle's sparse-LU trajectory
(tests/golden/oracle_flow.json) at the five checkpoints, both lab meshes: the noise floor the 3e-9 of tests/test_deploy_gpu.py
sits on."""
import json, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_deploy_gpu import _config
from meshdqn_amd.env import Env2DAirfoil
G = os.path.join(R, "tests", "golden")
for mesh in ("ys930", "ah93w145"):
    flow = json.load(open(os.path.join(G, "oracle_flow.json")))[mesh]["steps"]
    e = Env2DAirfoil(_config(mesh))
    for k in range(5):
        g = flow[str(1000 * (k + 1))]
        print(mesh, 1000 * (k + 1), f"drag {abs(e.gt_drag[k] - g['drag']) / abs(g['drag']):.2e} lift {abs(e.gt_lift[k] - g['lift']) / abs(g['lift']):.2e}")
