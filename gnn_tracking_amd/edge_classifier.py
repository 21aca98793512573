"""Edge classifier ``ECForGraphTCN`` on the fused HIP kernels.

Reference: models/edge_classifier.py:15-121.  Same constructor keywords, ``hparams``,
``state_dict`` keys (``ec_node_encoder``, ``ec_edge_encoder``, ``ec_resin``, ``W``),
``latent_dim`` and output dict ``{"W", "node_embedding", "edge_embedding"}``, so the
YAML ``class_path`` swap is the whole integration.

The whole stack runs with edges in target-sorted order: the edge encoder gathers
``edge_attr`` through the CSR permutation while reading it, every intermediate edge
embedding stays in CSR order, and ``W`` / ``edge_embedding`` are handed out as
``edge_order.EdgeOrdered``: tensors that behave as if they were in the caller's edge order
and pay for the scatter only when something other than this package's losses reads them.
"""

from __future__ import annotations

import math

import torch
from torch import Tensor, nn

from . import _capi, locality, ops, ops_bf16, precision
from .edge_order import EdgeOrdered, NodeOrdered
from .hparams import HyperparametersMixin, assert_feat_dim
from .mlp import MLP
from .resin import ResIN


def _n_events(data) -> int:
    """Events of a collated batch where the container says so without a device read (``ptr`` of ``collate`` /
    PyG's ``Batch``: one more entry than events); 0 = not stated."""
    ptr = getattr(data, "ptr", None)
    return int(ptr.numel()) - 1 if isinstance(ptr, Tensor) and ptr.numel() > 1 else 0


class ECForGraphTCN(nn.Module, HyperparametersMixin):
    def __init__(self, *, node_indim: int, edge_indim: int, interaction_node_dim: int = 5,
                 interaction_edge_dim: int = 4, hidden_dim: int | float | None = None,
                 L_ec: int = 3, alpha: float = 0.5, residual_type="skip1",
                 use_intermediate_edge_embeddings: bool = True, use_node_embedding: bool = True,
                 residual_kwargs: dict | None = None):
        """Edge classification step of the graph track condensation network.

        Args:
            node_indim: node feature dim
            edge_indim: edge feature dim
            interaction_node_dim: node dimension inside the interaction networks
            interaction_edge_dim: edge dimension inside the interaction networks
            hidden_dim: width of all hidden layers; ``None`` = per MLP max(in, out)
            L_ec: message passing depth
            alpha: strength of the residual connection
            residual_type: 'skip1', 'skip2' or 'skip_top'
            use_intermediate_edge_embeddings: feed the edge embeddings of all levels
                (not only the last) to the final MLP
            use_node_embedding: feed the endpoint node embeddings to the final MLP
            residual_kwargs: keyword arguments passed on to ``ResIN``
        """
        super().__init__()
        self.save_hyperparameters()
        residual_kwargs = dict(residual_kwargs or {})
        residual_kwargs["collect_hidden_edge_embeds"] = use_intermediate_edge_embeddings
        self.relu = nn.ReLU()
        self.ec_node_encoder = MLP(node_indim, interaction_node_dim, hidden_dim=hidden_dim, L=2,
                                   bias=False)
        self.ec_edge_encoder = MLP(edge_indim, interaction_edge_dim, hidden_dim=hidden_dim, L=2,
                                   bias=False)
        self.ec_resin = ResIN(node_dim=interaction_node_dim, edge_dim=interaction_edge_dim,
                              object_hidden_dim=hidden_dim, relational_hidden_dim=hidden_dim,
                              alpha=alpha, n_layers=L_ec, residual_type=residual_type,
                              residual_kwargs=residual_kwargs)
        w_input_dim = interaction_edge_dim
        if use_intermediate_edge_embeddings:
            w_input_dim = self.ec_resin.concat_edge_embeddings_length
        if use_node_embedding:
            w_input_dim += interaction_node_dim * 2
        self.W = MLP(input_size=w_input_dim, output_size=1, hidden_dim=hidden_dim, L=3)
        #: node, edge dim of the space before the final MLP
        self.latent_dim = (interaction_node_dim, interaction_edge_dim)

    def forward(self, data) -> dict[str, Tensor]:
        """``data`` exposes ``x``, ``edge_index``, ``edge_attr`` (PyG ``Data`` or any
        attribute bag).  Returns

        * ``W``: edge weights in (0.001, 0.999), order of ``edge_index``
        * ``node_embedding``: last node embedding
        * ``edge_embedding``: last edge embedding, order of ``edge_index``
        """
        x, edge_index, edge_attr = data.x, data.edge_index, data.edge_attr
        assert_feat_dim(x, self.hparams.node_indim)
        assert_feat_dim(edge_attr, self.hparams.edge_indim)
        bf16 = precision.use_bf16()
        # two per-edge inputs ride along into CSR order inside the graph-index build instead of being
        # gathered through the permutation afterwards: the dataset's 1-byte labels (for this package's
        # losses, which find them on the index) and, in bf16 storage, the four fp32 edge features
        y = getattr(data, "y", None)
        # node order (locality.py): the index is built in a renumbering that sorts every event's hits by
        # azimuth, x is gathered through it below and the node embedding handed back in the caller's order
        # (None as well when io.renumber_nodes / GraphDataset(renumber=True) ordered these events when they were read)
        col = locality.order_column(data)
        batch = getattr(data, "batch", None)
        # (a batch collated by io.ResidentDataset comes with its index placed from the per-event ones)
        gi = ops.placed_graph_index(edge_index, x.shape[0])
        if gi is None:
            gi = ops.graph_index(edge_index, x.shape[0],
                                 carry_label=y if isinstance(y, Tensor) and self.training else None,
                                 carry_rows=edge_attr if bf16 and edge_attr.shape[1] == 4 else None,
                                 order_by=None if col is None else (x, col, batch if isinstance(batch, Tensor) else None,
                                                                    _n_events(data)))
        E = gi.n_edges
        nperm = gi.node_perm

        if bf16:
            # bf16 storage: the dataset's fp32 features are converted once (edge_attr permuted
            # into CSR order in the same pass); everything downstream is bf16 rows, W is fp32
            h = self.ec_node_encoder.fused([ops.Seg(ops_bf16.to_rows16(x, nperm))], epilogue=_capi.EPI_RELU)
            ea_csr = ops.carried_rows(gi, edge_attr)
            if ea_csr is None:
                ea_csr = ops_bf16.to_rows16(edge_attr, gi.perm)
            e = self.ec_edge_encoder.fused([ops.Seg(ea_csr)], n_rows=E, epilogue=_capi.EPI_RELU)
        else:
            h = self.ec_node_encoder.fused([ops.Seg(x) if nperm is None else ops.Seg(x, nperm, False, "perm")],
                                           n_rows=x.shape[0], epilogue=_capi.EPI_RELU)
            e = self.ec_edge_encoder.fused([ops.Seg(edge_attr, gi.perm, False, "perm")],
                                           n_rows=E, epilogue=_capi.EPI_RELU)
        h, e, es = self.ec_resin.forward_csr(gi, h, e)

        segs = []
        if self.hparams.use_node_embedding:
            segs += [ops.Seg(h, gi.src, False, ("src", gi)), ops.Seg(h, gi.tgt, False, ("tgt", gi))]
        if self.hparams.use_intermediate_edge_embeddings:
            # (every embedding but the last is also read by the next interaction network: the head's
            #  gradient reaches its producer's backward kernel as an extra term, ops_bf16.grad_tap)
            segs += [ops.Seg(ops_bf16.grad_tap(t) if i + 1 < len(es) else t) for i, t in enumerate(es)]
        else:
            segs.append(ops.Seg(e))
        eps = 0.001
        # W and the edge embedding stay in CSR order; they present themselves in edge_index
        # order (edge_order.EdgeOrdered: the scatter happens only if something other than
        # this package's losses looks at the values)
        w = self.W.fused(segs, n_rows=E, epilogue=_capi.EPI_SIGMOID, ca=eps, cb=1 - 2 * eps)
        return {
            "W": EdgeOrdered(w.squeeze(), gi),
            "node_embedding": h if nperm is None else NodeOrdered(h, gi),
            "edge_embedding": EdgeOrdered(e, gi),
        }


class PerfectEdgeClassification(nn.Module, HyperparametersMixin):
    def __init__(self, tpr=1.0, tnr=1.0, false_below_pt=0.0):
        """Truth-based edge classifier (models/edge_classifier.py:124-163): ``W = y`` with an
        optional rate of flipped true / false edges and a pt cut.  No kernel of its own - it
        exists so that ``PerfectECGraphTCN`` configurations run unchanged."""
        super().__init__()
        self.save_hyperparameters()
        assert 0.0 <= tpr <= 1.0
        assert 0.0 <= tnr <= 1.0
        self.tpr, self.tnr, self.false_below_pt = tpr, tnr, false_below_pt

    def forward(self, data) -> dict[str, Tensor]:
        """Same random draws in the same order as the reference (one uniform per true edge,
        then one per edge that is false after the first step), written as mask algebra."""
        truth = data.y.bool()
        w = truth
        exact = lambda p: math.isclose(p, 1.0, rel_tol=1e-5, abs_tol=1e-8)  # noqa: E731
        none = torch.zeros_like(truth)
        if not exact(self.tpr):
            kept = torch.rand(int(truth.sum()), device=truth.device) <= self.tpr
            w = none.masked_scatter(truth, kept)
        if not exact(self.tnr):
            negative = ~w
            promoted = torch.rand(int(negative.sum()), device=truth.device) > self.tnr
            w = w | none.masked_scatter(negative, promoted)
        if self.false_below_pt > 0.0:
            w = w.masked_fill(data.pt < self.false_below_pt, False)
        return {"W": w.to(torch.float32)}
