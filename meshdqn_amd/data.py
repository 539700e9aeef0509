"""Minimal stand-ins for `torch_geometric.data.Data` / `Batch` / `DataLoader`
(as used by `Env2DAirfoil.get_state`, Env2DAirfoil.py:290, and the trainer, airfoil_dqn.py:256,268)."""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, batch=None, **kw):
        self.x, self.edge_index, self.edge_attr, self.batch = x, edge_index, edge_attr, batch
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    def to(self, device):
        out = Data(edge_attr=self.edge_attr)
        out.x = self.x.to(device) if self.x is not None else None
        out.edge_index = self.edge_index.to(device) if self.edge_index is not None else None
        out.batch = self.batch.to(device) if self.batch is not None else None
        for k, v in self.__dict__.items():
            if k not in ("x", "edge_index", "edge_attr", "batch"):
                setattr(out, k, v)
        return out

    def __repr__(self):
        e = 0 if self.edge_index is None else self.edge_index.shape[1]
        return f"Data(x={tuple(self.x.shape)}, edge_index=(2, {e}))"


class Batch(Data):
    """Concatenation of graphs: x stacked, edge_index offset by the cumulative node counts,
    `batch` = graph id per node (PyG `Batch.from_data_list`)."""

    @staticmethod
    def from_data_list(graphs: List[Data]) -> "Batch":
        xs, eis, bs = [], [], []
        off = 0
        for g, d in enumerate(graphs):
            n = d.x.shape[0]
            xs.append(d.x)
            ei = d.edge_index if d.edge_index is not None and d.edge_index.numel() else torch.zeros((2, 0), dtype=torch.long, device=d.x.device)
            eis.append(ei.reshape(2, -1) + off)
            bs.append(torch.full((n,), g, dtype=torch.long, device=d.x.device))
            off += n
        out = Batch(x=torch.cat(xs), edge_index=torch.cat(eis, dim=1), batch=torch.cat(bs))
        out.num_graphs = len(graphs)
        return out


class DataLoader:
    """`DataLoader(list_of_Data, batch_size)` yielding `Batch`es in order (no shuffling, as the reference uses it)."""

    def __init__(self, dataset: Iterable[Data], batch_size: int = 1):
        self.dataset = list(dataset)
        self.batch_size = int(batch_size)

    def __iter__(self):
        for i in range(0, len(self.dataset), self.batch_size):
            yield Batch.from_data_list(self.dataset[i:i + self.batch_size])

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size
