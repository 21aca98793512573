"""Graph construction from an embedding space on the HIP kNN kernel.

Reference: models/graph_construction.py:222-413 (``knn_with_max_radius``,
``MLGraphConstruction``).  The kNN search replaces ``torch_cluster.knn_graph``; edge
labels and edge features are built by two small gather kernels.  The optional embedding
network ``ml`` and edge filter ``ec`` are ordinary modules supplied by the caller.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from . import ops
from .data import Data
from .hparams import HyperparametersMixin, obj_from_or_to_hparams


def knn_with_max_radius(x: Tensor, k: int, max_radius: float | None = None) -> Tensor:
    """kNN graph that drops edges longer than ``max_radius`` (graph_construction.py:222-237).

    Returns an int64 edge index ``[2, M]``: row 0 = neighbour, row 1 = query."""
    return ops.knn_graph(x, k, max_radius)


def knn_scan(x: Tensor, ks, max_radius: float | None = None) -> dict[int, Tensor]:
    """Edge lists of ``knn_with_max_radius(x, k, max_radius)`` for every ``k`` in ``ks`` from
    one neighbour search at ``max(ks)``: the device part of ``GraphConstructionKNNScanner``
    (graph_construction/k_scanner.py:203-285, which searches once per k; its figures of
    merit are CPU tracking metrics and stay with the caller)."""
    return ops.knn_scan(x, ks, max_radius)


def _freeze_if(module, freeze: bool):
    if module is not None and freeze:
        for p in module.parameters():
            p.requires_grad = False
    return module


class MLGraphConstruction(nn.Module, HyperparametersMixin):
    def __init__(self, ml: torch.nn.Module | None = None, *, ec: torch.nn.Module | None = None,
                 max_radius: float = 1, max_num_neighbors: int = 256,
                 use_embedding_features=False, ratio_of_false=None, build_edge_features=True,
                 ec_threshold=None, ml_freeze: bool = True, ec_freeze: bool = True,
                 embedding_slice: tuple[int | None, int | None] = (None, None)):
        """Builds a graph from embedding space (graph_construction.py:240-413).

        Args:
            ml: metric-learning embedding module (``forward(data) -> {"H": ...}``); if
                None the node features (``embedding_slice``) are the embedding
            ec: edge filter applied to the built edges (needs ``ec_threshold``)
            max_radius: maximum edge length in embedding space
            max_num_neighbors: k of the kNN search
            use_embedding_features: prepend the embedding to the node features
            ratio_of_false: in training, keep at most this many false edges per true edge
            build_edge_features: edge features ``[x_j - x_i, x_j + x_i]``
            ec_threshold: threshold of the edge filter
            embedding_slice: slice of the node features used as embedding when ``ml`` is None
        """
        super().__init__()
        self.save_hyperparameters(ignore=["ml", "ec"])
        self._ml = _freeze_if(obj_from_or_to_hparams(self, "ml", ml), ml_freeze)
        self._ef = _freeze_if(obj_from_or_to_hparams(self, "ec", ec), ec_freeze)
        if self._ef is not None and ec_threshold is None:
            raise ValueError("ec_threshold must be set if ec/ef is not None")
        if self._ml is None and use_embedding_features:
            raise ValueError("use_embedding_features requires ml to be not None")
        if self._ml is not None and tuple(embedding_slice) != (None, None):
            raise ValueError("embedding_slice requires ml to be None")

    @property
    def out_dim(self) -> tuple[int, int]:
        if self._ml is None:
            raise RuntimeError("Cannot infer output dimension without ML model")
        node_dim: int = self._ml.in_dim
        if self.hparams.use_embedding_features:
            node_dim += self._ml.out_dim
        return node_dim, (2 * node_dim if self.hparams.build_edge_features else 0)

    def forward(self, data) -> Data:
        if not hasattr(data, "true_edge_index"):
            data.true_edge_index = data.edge_index
        if self._ml is not None:
            mo = self._ml(data)
            emb = mo["H"]
        else:
            s = self.hparams.embedding_slice
            emb = data.x[:, s[0]:s[1]]
        edge_index = knn_with_max_radius(emb, max_radius=self.hparams.max_radius,
                                         k=self.hparams.max_num_neighbors)
        y = ops.edge_labels(data.particle_id, edge_index)
        if self._ml is None or not self.hparams.use_embedding_features:
            x = data.x
        else:
            x = torch.cat((mo["H"], data.x), dim=1)
        if self.hparams.ratio_of_false and self.training:
            yb = y.bool()
            n_keep = int(int(yb.sum()) * self.hparams.ratio_of_false)
            false_edges = edge_index[:, ~yb][:, :n_keep]
            true_edges = edge_index[:, yb]
            edge_index = torch.cat((false_edges, true_edges), dim=1).contiguous()
            y = torch.cat((torch.zeros(false_edges.shape[1], dtype=y.dtype, device=y.device),
                           torch.ones(true_edges.shape[1], dtype=y.dtype, device=y.device)))
        edge_attr = None
        if self.hparams.build_edge_features:
            edge_attr = ops.edge_features(x, edge_index)
        if self._ef is not None:
            w = self._ef(Data(x=x, edge_index=edge_index, edge_attr=edge_attr))["W"]
            mask = w > self.hparams.ec_threshold
            edge_index = edge_index[:, mask].contiguous()
            y = y[mask]
            edge_attr = edge_attr[mask]
        return Data(x=x, edge_index=edge_index, true_edges=data.true_edge_index, y=y.long(),
                    pt=data.pt, particle_id=data.particle_id,
                    sector=getattr(data, "sector", None),
                    reconstructable=data.reconstructable, edge_attr=edge_attr, eta=data.eta,
                    layer=getattr(data, "layer", None))


class MLPCTransformer(nn.Module, HyperparametersMixin):
    def __init__(self, model: nn.Module, *, original_features: bool = False, freeze: bool = True):
        """Transform a point cloud with a metric-learning model (models/graph_construction.py:
        422-481): ``data.x`` becomes the latent space ``H`` (optionally followed by the original
        features).  As in the reference the ``Data`` object is modified in place.  (The
        ``from_ml_chkpt`` constructors read Lightning checkpoints: control plane, not here.)"""
        super().__init__()
        self._ml = _freeze_if(obj_from_or_to_hparams(self, "ml", model), freeze)
        self.save_hyperparameters(ignore=["model"])

    def forward(self, data) -> Data:
        out = self._ml(data)
        if self.hparams.original_features:
            data.x = torch.cat((out["H"], data.x), dim=1)
        else:
            data.x = out["H"]
        return data
