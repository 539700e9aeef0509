"""GPU parity: fused HIP forward of the Q-networks (C ABI mdq_gcn_forward) vs the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(rng, n, e, f):
    from meshdqn_amd.data import Data
    x = torch.from_numpy(rng.standard_normal((n, f))).float()
    ei = torch.from_numpy(rng.integers(0, n, size=(2, e))).long()
    return Data(x=x, edge_index=ei)


@pytest.mark.parametrize("cls,kw,feat", [("NodeRemovalNet", dict(output_dim=181, conv_width=128, topk=0.1), 17),
                                          ("AirfoilGCNN", dict(conv_width=64), 17),
                                          ("AirfoilGCNN", dict(conv_width=128), 17)])   # (GCNConv at the width of the node-per-lane form)
def test_fused_forward_matches_oracle(lib_built, cls, kw, feat):
    from meshdqn_amd import airfoilgcnn as prod
    from meshdqn_amd.data import Batch
    from oracle import gcn as ora
    rng = np.random.default_rng(11)
    net_p = getattr(prod, cls)(**kw)
    net_o = getattr(ora, cls)(**kw)
    if cls == "NodeRemovalNet":
        net_p.set_num_nodes(feat)
        net_o.set_num_nodes(feat)
    sd = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)) * 0.3).float() for k, v in net_p.state_dict().items()}
    net_p.load_state_dict(sd)
    net_o.load_state_dict(sd)
    net_p = net_p.cuda()
    sizes = [(180, 372), (180, 495), (37, 60), (180, 0), (64, 300), (180, 420)] + [(180, 400)] * 34
    if cls == "AirfoilGCNN" and kw["conv_width"] == 128:      # (its pooling ratio keeps more rows: 180-node graphs exceed the LDS)
        sizes = [(100, 250), (64, 150), (37, 60), (100, 0), (90, 300), (17, 30)] + [(100, 220)] * 10
    graphs = [_graph(rng, n, e, feat) for n, e in sizes]
    batch = Batch.from_data_list(graphs)
    with torch.no_grad():
        yo = net_o(batch)
        yt = net_p(batch.to("cuda")).cpu()       # torch path on the GPU
        yf = net_p.forward_fused(batch.to("cuda")).cpu()  # fused HIP path
    assert yf.shape == yo.shape
    scale = yo.abs().max().item()
    assert (yt - yo).abs().max().item() < 2e-4 * max(scale, 1e-3)
    assert (yf - yo).abs().max().item() < 2e-4 * max(scale, 1e-3)
    if cls == "NodeRemovalNet":
        # greedy actions agree wherever the top-2 margin is not at round-off level
        top2 = yo.topk(2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert (yf.argmax(1)[clear] == yo.argmax(1)[clear]).all()
        # a bare Data (batch=None) works too
        y1 = net_p.forward_fused(graphs[0].to("cuda")).cpu()
        assert (y1[0] - yo[0]).abs().max().item() < 2e-4
