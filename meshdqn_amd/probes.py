"""`DragProbe` / `LiftProbe` - the reference's probes (probes.py:13-50) on the HIP facet-integral kernel.

    probe = DragProbe(mu, n, ds, tags=[1]);  value = probe.sample(u, p)

`n` (the UFL FacetNormal of the reference) is not needed: the kernel derives the outward normal
of every tagged facet from the cell geometry.  `ds` carries the surface measure in the
reference; here it is the object that owns the mesh batch (the `FlowSolver`).
"""
from __future__ import annotations

import torch


class _ForceProbe(object):
    component = 0

    def __init__(self, mu, n, ds, tags, flow_dir=None):
        self.mu = mu
        self.n = n
        self.ds = ds
        self.tags = list(tags)
        if self.tags != [1]:
            raise NotImplementedError("only the airfoil surface (tag 1) is integrated, as in the reference")
        self.flow_dir = flow_dir
        self.dim = 2

    def _forces(self, u, p):
        batch = self.ds.probe_batch()
        N2, NV = batch.N2, batch.cap["NV"]
        ub = torch.zeros((1, 1, N2, 2), dtype=torch.float64, device=batch.device)
        pb = torch.zeros((1, 1, NV), dtype=torch.float64, device=batch.device)
        ub[0, 0, :u.data.shape[0]] = u.data
        pb[0, 0, :p.data.shape[0]] = p.data
        drag, lift = batch.probe_forces(ub, pb)
        return drag[0, 0].item(), lift[0, 0].item()

    def sample(self, u, p):
        return self._forces(u, p)[self.component]


class DragProbe(_ForceProbe):
    """Integral probe of drag over the tagged exterior surface (probes.py:13-31)."""
    component = 0


class LiftProbe(_ForceProbe):
    """Integral probe of lift over the tagged exterior surface (probes.py:33-50)."""
    component = 1
