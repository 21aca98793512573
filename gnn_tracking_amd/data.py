"""Minimal graph container + collate with the part of the PyG ``Data`` /
``DataLoader`` behaviour the reference's hot path relies on.

The modules of this package accept ANY object exposing ``.x / .edge_index /
.edge_attr / ...`` (a real ``torch_geometric.data.Data`` works unchanged); this class
exists so the package, its tests and the bench run without PyG installed.
"""

from __future__ import annotations

import copy
from typing import Iterable

import torch
from torch import Tensor


class Data:
    """Attribute bag.  Node-level attributes have ``num_nodes`` rows, edge-level ones
    (``edge_attr``, ``y``, names starting with ``edge_``) have ``num_edges`` rows."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def keys(self) -> list[str]:
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def __contains__(self, k) -> bool:
        return k in self.__dict__

    def __getattr__(self, k):
        # PyG declares these as properties that are None when the attribute was never set
        # (training/tc.py:61 reads ``data.batch`` of a graph built by MLGraphConstruction)
        if k in ("batch", "pos", "edge_weight", "edge_attr", "y", "x", "edge_index"):
            return None
        raise AttributeError(k)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    @property
    def num_node_features(self) -> int:
        return int(self.x.shape[1])

    @property
    def num_edge_features(self) -> int:
        return int(self.edge_attr.shape[1])

    def _map(self, fn):
        out = copy.copy(self)
        for k in self.keys():
            v = getattr(self, k)
            if torch.is_tensor(v):
                setattr(out, k, fn(v))
        return out

    def to(self, device, **kw):
        return self._map(lambda t: t.to(device, **kw))

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def detach(self):
        return self._map(lambda t: t.detach())

    # -- PyG ``Data.edge_subgraph`` / ``Data.subgraph`` as the reference uses them
    # (models/track_condensation_networks.py:252,259): every edge-level attribute is
    # filtered by the edge mask; ``subgraph`` keeps a node subset, filters node-level
    # attributes, drops edges touching removed nodes and relabels ``edge_index``.
    def edge_subgraph(self, mask: Tensor) -> "Data":
        out = copy.copy(self)
        for k in self.keys():
            v = getattr(self, k)
            if k == "edge_index":
                out.edge_index = v[:, mask]
            elif self.is_edge_attr(k):
                setattr(out, k, v[mask])
        return out

    def subgraph(self, subset: Tensor) -> "Data":
        n = self.num_nodes
        dev = self.edge_index.device
        if subset.dtype == torch.bool:
            node_mask = subset
        else:
            node_mask = torch.zeros(n, dtype=torch.bool, device=dev)
            node_mask[subset] = True
        relabel = torch.full((n,), -1, dtype=torch.long, device=dev)
        relabel[node_mask] = torch.arange(int(node_mask.sum()), device=dev)
        ei = self.edge_index
        emask = node_mask[ei[0]] & node_mask[ei[1]]
        out = copy.copy(self)
        for k in self.keys():
            v = getattr(self, k)
            if k == "edge_index":
                out.edge_index = relabel[ei[:, emask]]
            elif self.is_edge_attr(k):
                setattr(out, k, v[emask])
            elif self.is_node_attr(k):
                setattr(out, k, v[node_mask])
        return out

    def is_edge_attr(self, key: str) -> bool:
        if key == "edge_index" or "index" in key:
            return False
        v = getattr(self, key)
        return (torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == self.num_edges
                and (key.startswith("edge_") or key in ("y", "ec_edge_embedding")))

    def is_node_attr(self, key: str) -> bool:
        v = getattr(self, key)
        return (torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == self.num_nodes
                and "index" not in key and not self.is_edge_attr(key))


def collate(graphs: Iterable[Data]) -> Data:
    """Concatenate graphs into one disjoint graph the way PyG's ``Batch`` does:
    node-/edge-level tensors are concatenated along dim 0, every attribute whose name
    contains ``index`` is concatenated along dim -1 after adding the cumulative node
    count, and ``batch`` (graph id per node) / ``ptr`` (node offsets) are added."""
    graphs = list(graphs)
    if not graphs:
        raise ValueError("collate: empty list")
    out = Data()
    offs = [0]
    for g in graphs:
        offs.append(offs[-1] + g.num_nodes)
    for k in graphs[0].keys():
        vals = [getattr(g, k) for g in graphs]
        if not torch.is_tensor(vals[0]):
            setattr(out, k, vals)
        elif "index" in k:
            # (offset while copying: one read and one write of every id instead of a shifted temporary and its copy)
            cat = torch.empty(*vals[0].shape[:-1], sum(int(v.shape[-1]) for v in vals), dtype=vals[0].dtype,
                              device=vals[0].device)
            a = 0
            for v, o in zip(vals, offs):
                if v.shape[:-1] != vals[0].shape[:-1] or v.dtype != vals[0].dtype:   # (what torch.cat would refuse)
                    raise RuntimeError(f"collate: '{k}' of the events disagrees in shape / dtype: {tuple(vals[0].shape)} "
                                       f"{vals[0].dtype} against {tuple(v.shape)} {v.dtype}")
                torch.add(v, o, out=cat[..., a:a + v.shape[-1]])
                a += int(v.shape[-1])
            setattr(out, k, cat)
        elif vals[0].dim() == 0:
            setattr(out, k, torch.stack(vals))
        else:
            setattr(out, k, torch.cat(vals, dim=0))
    dev = graphs[0].x.device
    out.batch = torch.cat([torch.full((g.num_nodes,), i, dtype=torch.long, device=dev)
                           for i, g in enumerate(graphs)])
    out.ptr = torch.tensor(offs, dtype=torch.long, device=dev)
    return out
