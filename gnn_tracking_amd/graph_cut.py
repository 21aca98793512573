"""Threshold cut and orphan-node masking of ``ModularGraphTCN`` on the device
(models/track_condensation_networks.py:251-262).

``threshold_compact`` / ``connected_nodes`` are the C-ABI stream compactions
(csrc/compact.hip); ``edge_cut`` / ``drop_orphans`` apply their index lists to a graph
container the way PyG's ``Data.edge_subgraph`` / ``Data.subgraph`` do: every edge-level
(node-level) attribute keeps the selected rows in ascending order, ``edge_index`` is
relabelled.  One host read of the counts per cut (the reference's boolean indexing
synchronises once per attribute).
"""

from __future__ import annotations

import copy

import torch
from torch import Tensor

from . import _capi, ops
from .data import Data


def _workspace(lib, n: int, like: Tensor) -> Tensor:
    return torch.empty(max(int(lib.gnntrk_compact_workspace_bytes(int(n))), 256), dtype=torch.uint8,
                       device=like.device)


def threshold_compact(w: Tensor, threshold: float) -> tuple[Tensor, Tensor]:
    """``mask = w > threshold`` (bool ``[n]``) and the ascending positions of the kept
    entries (int32 ``[n_kept]``): gnntrk_threshold_compact."""
    _capi.require_device(w)
    lib = _capi.load()
    w = w.detach().reshape(-1).to(torch.float32).contiguous()
    n = int(w.shape[0])
    mask = torch.empty(n, dtype=torch.uint8, device=w.device)
    idx = torch.empty(n, dtype=torch.int32, device=w.device)
    n_out = torch.empty(1, dtype=torch.int64, device=w.device)
    ws = _workspace(lib, n, w)
    _capi.check(lib.gnntrk_threshold_compact(ops._p(w), n, float(threshold), ops._p(mask), ops._p(idx),
                                             ops._p(n_out), ops._p(ws), ws.numel(), ops._stream(w)), lib)
    return mask.bool(), idx[:int(n_out.item())]


def connected_nodes(edge_index: Tensor, num_nodes: int) -> tuple[Tensor, Tensor, Tensor]:
    """Nodes that are an endpoint of an edge: ``(hit_mask bool [N], node_idx int32 [n_conn]
    ascending, edge_index relabelled to the compacted node ids)``: gnntrk_connected_nodes."""
    _capi.require_device(edge_index)
    lib = _capi.load()
    ei = edge_index.to(torch.int64).contiguous()
    E, N = int(ei.shape[1]), int(num_nodes)
    dev = ei.device
    hit = torch.empty(N, dtype=torch.uint8, device=dev)
    node_idx = torch.empty(N, dtype=torch.int32, device=dev)
    newid = torch.empty(N, dtype=torch.int32, device=dev)
    n_out = torch.empty(2, dtype=torch.int64, device=dev)
    ei_out = torch.empty_like(ei)
    ws = _workspace(lib, N, ei)
    _capi.check(lib.gnntrk_connected_nodes(ops._p(ei), E, N, ops._p(hit), ops._p(node_idx), ops._p(newid),
                                           ops._p(n_out), ops._p(ei_out), ops._p(ws), ws.numel(),
                                           ops._stream(ei)), lib)
    n_conn, bad = (int(v) for v in n_out.tolist())
    if bad:
        raise ValueError("edge_index contains node ids outside [0, num_nodes)")
    return hit.bool(), node_idx[:n_conn], ei_out


def _take(v: Tensor, idx: Tensor) -> Tensor:
    return v.index_select(0, idx)


def edge_cut(data, w: Tensor, threshold: float, lazy: tuple = (), only=None):
    """``mask = w > threshold; data.edge_subgraph(mask)`` -> ``(data', mask)``.

    ``only``: if given, the edge attributes (besides ``edge_index``) the caller will read from
    ``data'``; the others come back as ``None`` instead of being copied (``ModularGraphTCN`` keeps the
    cut graph to itself and reads two or three of them).

    ``lazy``: names of edge attributes the caller will read through a fused row gather instead
    (``ModularGraphTCN`` feeds ``edge_attr`` to the next encoder that way: the kept rows of a
    112-byte-per-edge tensor are then never copied).  Those attributes come back as ``None`` and
    ``data'._lazy_rows[name] = (uncut tensor, int32 kept-edge index)``; only honoured for this
    package's ``Data`` and for tensors that do not require a gradient."""
    mask, idx = threshold_compact(w, threshold)
    if not isinstance(data, Data):  # a foreign container (e.g. PyG): its own edge_subgraph
        return data.edge_subgraph(mask), mask
    idx32 = idx
    idx = idx.long()
    out = copy.copy(data)
    lazy_rows = {}
    for k in data.keys():
        v = getattr(data, k)
        if k == "edge_index":
            out.edge_index = v.index_select(1, idx)
        elif data.is_edge_attr(k):
            if only is not None and k not in only:
                setattr(out, k, None)
            elif k in lazy and torch.is_tensor(v) and v.dim() == 2 and not v.requires_grad:
                lazy_rows[k] = (v, idx32)
                setattr(out, k, None)
            else:
                setattr(out, k, _take(v, idx))
    out._lazy_rows = lazy_rows
    return out, mask


def drop_orphans(data):
    """``connected = data.edge_index.flatten().unique(); hit_mask = index_to_mask(connected);
    data.subgraph(connected)`` -> ``(data', hit_mask)``."""
    hit, node_idx, ei = connected_nodes(data.edge_index, data.num_nodes)
    if not isinstance(data, Data):
        return data.subgraph(node_idx.long()), hit
    node_idx = node_idx.long()
    out = copy.copy(data)
    for k in data.keys():
        v = getattr(data, k)
        if k == "edge_index":
            out.edge_index = ei
        elif data.is_node_attr(k):
            setattr(out, k, _take(v, node_idx))
    return out, hit
