"""Node masks (reference: utils/graph_masks.py:7-36)."""

from __future__ import annotations

import torch
from torch import Tensor

from . import _capi, ops


def get_good_node_mask_tensors(*, pt: Tensor, particle_id: Tensor, reconstructable: Tensor,
                               eta: Tensor, pt_thld: float = 0.9, max_eta: float = 4.0) -> Tensor:
    """``(pt > pt_thld) & (particle_id > 0) & (reconstructable > 0) & (|eta| < max_eta)``
    as a bool tensor (graph_masks.py:19-28), one fused kernel."""
    _capi.require_device(pt, particle_id, reconstructable, eta)
    lib = _capi.load()
    n = int(pt.shape[0])
    for name, t in (("particle_id", particle_id), ("reconstructable", reconstructable), ("eta", eta)):
        # (the reference's elementwise & raises on a mismatch too, e.g. an eta that was not
        # sliced by ec_hit_mask, oc.py:207-213)
        if t.dim() != 1 or int(t.shape[0]) != n or pt.dim() != 1:
            raise RuntimeError(f"get_good_node_mask_tensors: {name} has shape {tuple(t.shape)}, "
                               f"pt has shape {tuple(pt.shape)}: the node attributes must be 1-D of one length")
    pt32 = pt.to(torch.float32).contiguous()
    pid = particle_id.to(torch.int64).contiguous()
    reco = reconstructable.to(torch.float32).contiguous()
    eta32 = eta.to(torch.float32).contiguous()
    mask = torch.empty(n, dtype=torch.uint8, device=pt.device)
    _capi.check(lib.gnntrk_good_node_mask(ops._p(pt32), ops._p(pid), ops._p(reco), ops._p(eta32), n,
                                          float(pt_thld), float(max_eta), ops._p(mask),
                                          ops._stream(pt32)), lib)
    return mask.bool()


def get_good_node_mask(data, *, pt_thld: float = 0.9, max_eta: float = 4.0) -> Tensor:
    """graph_masks.py:7-16."""
    return get_good_node_mask_tensors(pt=data.pt, particle_id=data.particle_id,
                                      reconstructable=data.reconstructable, eta=data.eta,
                                      pt_thld=pt_thld, max_eta=max_eta)


def get_edge_mask_from_node_mask(node_mask: Tensor, edge_index: Tensor) -> Tensor:
    """Edges whose both endpoints are in the node mask (graph_masks.py:31-36); index
    bookkeeping on bool tensors."""
    return node_mask[edge_index[0].long()] & node_mask[edge_index[1].long()]
