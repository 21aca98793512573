"""Metric-learning hinge loss for graph construction.

Reference: metrics/losses/metric_learning.py:14-178 (``GraphConstructionHingeEmbeddingLoss``).
Same constructor keywords, ``hparams`` and ``MultiLossFctReturn`` (``attractive``,
``repulsive``; extra metrics ``n_hits_oi``, ``n_edges_att``, ``n_edges_rep``).

The neighbour search - the O(N^2 D) part - is the HIP kNN kernel with a radius cut
(``gnntrk_knn_search``: the ``max_num_neighbors`` nearest hits inside ``r_emb``, which is
torch_cluster's ``radius_graph`` whenever the cap is not reached; with the cap reached
torch_cluster keeps an implementation-defined subset), run per event of ``batch``; the
hit-of-interest mask is ``gnntrk_good_node_mask``.  The two edge-list reductions that follow
(gather, norm, power, hinge, sum, normalisation) are one fused kernel each
(``ops_ml.hinge_terms`` -> ``gnntrk_hinge_forward``), with the reference's edge selection - hits of
interest at the first endpoint, different particles at the two ends - applied inside the kernel
instead of compacting the edge lists; their backward is ONE node-centric pass over the graph index
of the edge list (``gnntrk_hinge_backward``: no per-edge intermediate, no atomics).
"""

from __future__ import annotations

import torch
from torch import Tensor as T
from torch import nn

from . import ops
from .graph_masks import get_good_node_mask_tensors
from .hparams import HyperparametersMixin
from .losses_oc import MultiLossFctReturn


def radius_graph(x: T, r: float, batch: T | None = None, max_num_neighbors: int = 32) -> T:
    """Edges ``[2, M]`` (row 0 = neighbour, row 1 = centre) between hits of the same event
    closer than ``r`` (no self loops), at most ``max_num_neighbors`` nearest per centre."""
    xd = x.detach()
    if batch is None:
        return ops.knn_graph(xd, max_num_neighbors, r)
    # all events of the collated batch in ONE search (gnntrk_knn_search_batched): row offsets of
    # the events from `batch` (sorted, as PyG's collate produces it - checked on the device
    # together with the one unavoidable read of the edge count)
    b = batch.long()
    counts = torch.bincount(b)
    seg_ptr = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=x.device)
    seg_ptr[1:] = torch.cumsum(counts, 0)
    if b.numel() > 1 and bool((b[1:] < b[:-1]).any()):
        raise ValueError("radius_graph: `batch` must be sorted (PyG convention)")
    return ops.knn_graph(xd, max_num_neighbors, r, seg_ptr=seg_ptr)


class GraphConstructionHingeEmbeddingLoss(nn.Module, HyperparametersMixin):
    def __init__(self, *, lw_repulsive: float = 1.0, r_emb: float = 1.0, max_num_neighbors: int = 256,
                 pt_thld: float = 0.9, max_eta: float = 4.0, p_attr: float = 1.0, p_rep: float = 1.0,
                 rep_normalization: str = "n_hits_oi", rep_oi_only: bool = True):
        """Loss for graph construction using metric learning.

        Args:
            lw_repulsive: loss weight of the repulsive part
            r_emb: radius for edge construction
            max_num_neighbors: maximum number of neighbours in the radius graph
            pt_thld: pt threshold for particles of interest
            max_eta: maximum eta for particles of interest
            p_attr: power of the attraction term
            p_rep: power of the repulsion term
            rep_normalization: "n_rep_edges", "n_hits_oi" or "n_att_edges"
            rep_oi_only: only repulsion from hits of interest
        """
        super().__init__()
        self.save_hyperparameters()

    def _get_edges(self, *, x: T, batch: T, true_edge_index: T, mask: T, particle_id: T):
        near_edges = radius_graph(x, r=self.hparams.r_emb, batch=batch,
                                  max_num_neighbors=self.hparams.max_num_neighbors)
        rep_edges = near_edges[:, mask[near_edges[0]]] if self.hparams.rep_oi_only else near_edges
        rep_edges = rep_edges[:, particle_id[rep_edges[0]] != particle_id[rep_edges[1]]]
        att_edges = true_edge_index[:, mask[true_edge_index[0]]]
        return att_edges, rep_edges

    def forward(self, *, x: T, particle_id: T, batch: T, true_edge_index: T, pt: T, eta: T,
                reconstructable: T, **kwargs) -> MultiLossFctReturn:
        if true_edge_index is None:
            raise ValueError(
                "True_edge_index must be given and not be None. Are you trying to use this loss for "
                "OC training? In this case, double check that you are properly passing on the true edges.")
        hp = self.hparams
        mask = get_good_node_mask_tensors(pt=pt, particle_id=particle_id, reconstructable=reconstructable,
                                          eta=eta, pt_thld=hp.pt_thld, max_eta=hp.max_eta)
        n_hits_oi = mask.sum()
        if x.dtype == torch.float32:   # (other dtypes - the reference's float64 known-answer tests - below)
            return self._forward_kernels(x=x, particle_id=particle_id, batch=batch, true_edge_index=true_edge_index,
                                         mask=mask, n_hits_oi=n_hits_oi)
        att_edges, rep_edges = self._get_edges(x=x, batch=batch, true_edge_index=true_edge_index,
                                               mask=mask, particle_id=particle_id)
        eps = 1e-9
        dists_att = torch.linalg.norm(x[att_edges[0]] - x[att_edges[1]], dim=-1)
        v_att = torch.sum(torch.pow(dists_att, hp.p_attr)) / (att_edges.shape[1] + eps)
        dists_rep = torch.linalg.norm(x[rep_edges[0]] - x[rep_edges[1]], dim=-1)
        if hp.rep_normalization == "n_rep_edges":
            norm_rep = rep_edges.shape[1] + eps
        elif hp.rep_normalization == "n_hits_oi":
            norm_rep = n_hits_oi + eps
        elif hp.rep_normalization == "n_att_edges":
            norm_rep = att_edges.shape[1] + eps
        else:
            raise ValueError(f"Normalization {hp.rep_normalization} not recognized.")
        v_rep = torch.sum(torch.relu(hp.r_emb - torch.pow(dists_rep, hp.p_rep))) / norm_rep
        return MultiLossFctReturn(
            loss_dct={"attractive": v_att, "repulsive": v_rep},
            weight_dct={"attractive": 1.0, "repulsive": hp.lw_repulsive},
            extra_metrics={"n_hits_oi": n_hits_oi, "n_edges_att": att_edges.shape[1],
                           "n_edges_rep": rep_edges.shape[1]})

    def _forward_kernels(self, *, x: T, particle_id: T, batch: T, true_edge_index: T, mask: T, n_hits_oi: T):
        """fp32 embedding on the device: the kernels of csrc/hinge.hip.  The edge counts stay on the device
        (one-element tensors in ``extra_metrics``): nothing in the loss waits for the host."""
        from . import ops_ml
        hp = self.hparams
        if hp.rep_normalization not in ("n_rep_edges", "n_hits_oi", "n_att_edges"):
            raise ValueError(f"Normalization {hp.rep_normalization} not recognized.")
        near_edges = radius_graph(x, r=hp.r_emb, batch=batch, max_num_neighbors=hp.max_num_neighbors)
        v_att, n_att, _ = ops_ml.hinge_terms(x, true_edge_index, node_mask=mask, p=hp.p_attr, repulsive=False)
        norm = {"n_rep_edges": None, "n_hits_oi": n_hits_oi, "n_att_edges": n_att}[hp.rep_normalization]
        v_rep, n_rep, _ = ops_ml.hinge_terms(x, near_edges, node_mask=mask if hp.rep_oi_only else None,
                                             particle_id=particle_id, norm=norm, r_emb=hp.r_emb, p=hp.p_rep,
                                             repulsive=True)
        return MultiLossFctReturn(
            loss_dct={"attractive": v_att, "repulsive": v_rep},
            weight_dct={"attractive": 1.0, "repulsive": hp.lw_repulsive},
            extra_metrics={"n_hits_oi": n_hits_oi, "n_edges_att": n_att, "n_edges_rep": n_rep})
