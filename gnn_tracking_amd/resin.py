"""Stacks of interaction networks with residual node connections.

Reference: models/resin.py:17-295.  ``ResIN`` keeps the constructor keywords,
``hparams`` and ``state_dict`` keys (``network.layers.<l>.*``); the three residual
layouts (``skip1``, ``skip2``, ``skip_top``) are wiring over the same fused kernels:
input ReLUs are folded into the kernels' loads and the residual combination
``sqrt(a)*residue + sqrt(1-a)*delta`` into the object model's epilogue.

Every stack exposes ``forward_csr(gi, x, e_csr)`` (edge tensors in CSR order, used by
``ECForGraphTCN``) next to the reference's ``forward(x, edge_index, edge_attr)``.
"""

from __future__ import annotations

import math
from typing import Any

import torch
from torch import Tensor, nn

from . import ops
from .hparams import HyperparametersMixin
from .interaction_network import InteractionNetwork


def sqconvex_combination(*, delta: Tensor, residue: Tensor | None, alpha_residue: float) -> Tensor:
    """resin.py:29-42 (kept for API parity; the stacks below fuse it)."""
    if residue is None or math.isclose(alpha_residue, 0.0):
        return delta
    return ops.axpby(math.sqrt(alpha_residue), residue, math.sqrt(1 - alpha_residue), delta)


def _res(residue, alpha):
    """Residue to fuse, or None when the combination is the identity on delta."""
    return None if (residue is None or math.isclose(alpha, 0.0)) else residue


class ResidualNetwork(nn.Module):
    def __init__(self, layers: list[nn.Module], *, alpha: float = 0.5,
                 collect_hidden_edge_embeds: bool = False):
        """Sequence of IN layers with node residuals (resin.py:45-89).

        Args:
            layers: the interaction networks
            alpha: strength of the node residual connection
            collect_hidden_edge_embeds: also return the edge embeddings of all levels
        """
        super().__init__()
        self.layers = nn.ModuleList(layers)
        self._alpha = alpha
        self._collect_hidden_edge_embeds = collect_hidden_edge_embeds

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor):
        """Returns ``(node embedding, edge embedding, [edge embeddings of all levels
        incl. edge_attr] or None)`` - edge tensors in the order of ``edge_index``."""
        gi = ops.graph_index(edge_index, x.shape[0])
        e_csr = ops.permute_rows(edge_attr, gi.perm, scatter=False)
        x, e, es = self.forward_csr(gi, x, e_csr)
        back = lambda t: ops.permute_rows(t, gi.perm, scatter=True)  # noqa: E731
        if es is not None:
            es = [edge_attr] + [back(t) for t in es[1:]]
        return x, back(e), es

    def forward_csr(self, gi: ops.GraphIndex, x: Tensor, e_csr: Tensor):
        raise NotImplementedError


class Skip1ResidualNetwork(ResidualNetwork):
    """Residual connection between any two successive layers (resin.py:92-114)."""

    def forward_csr(self, gi, x, e):
        es = [e] if self._collect_hidden_edge_embeds else None
        for i, layer in enumerate(self.layers):
            x, e = layer.forward_csr(gi, x, e, relu_in=i > 0, residue=_res(x, self._alpha),
                                     alpha_residue=self._alpha)
            if es is not None:
                es.append(e)
        return x, e, es


class Skip2ResidualNetwork(ResidualNetwork):
    def __init__(self, layers: list[nn.Module], *, node_dim: int, edge_dim: int,
                 add_bn: bool = False, **kwargs):
        """Blocks of two layers joined by a residual (resin.py:117-175).  The reference
        walks ``pairwise(range(n))`` - overlapping pairs (0,1),(1,2),... - and so does
        this.  ``add_bn``: ``BatchNorm1d`` on the node and edge inputs of every layer
        (resin.py:143-151; torch's own batch norm between the fused kernels - its statistics
        are sums over rows, so the CSR edge order does not change them; note that it couples the
        events of a collated batch, exactly as in the reference)."""
        if len(layers) % 2 != 0:
            raise ValueError("Only even number of layers allowed at the moment")
        super().__init__(layers=layers, **kwargs)
        self._add_bn = bool(add_bn)
        self._node_batch_norms = nn.ModuleList(
            [nn.BatchNorm1d(node_dim) if add_bn else nn.Identity() for _ in layers])
        self._edge_batch_norms = nn.ModuleList(
            [nn.BatchNorm1d(edge_dim) if add_bn else nn.Identity() for _ in layers])

    def _bn(self, norms, i: int, t):
        if not self._add_bn:
            return t
        return norms[i](t.float()).to(t.dtype)

    def forward_csr(self, gi, x, e):
        es = [e] if self._collect_hidden_edge_embeds else None
        n = len(self.layers)
        for i0 in range(n - 1):
            i1 = i0 + 1
            hx, he = self.layers[i0].forward_csr(gi, self._bn(self._node_batch_norms, i0, x),
                                                 self._bn(self._edge_batch_norms, i0, e), relu_in=i0 > 0)
            x, e = self.layers[i1].forward_csr(gi, self._bn(self._node_batch_norms, i1, hx),
                                               self._bn(self._edge_batch_norms, i1, he), relu_in=True,
                                               residue=_res(x, self._alpha), alpha_residue=self._alpha)
            if es is not None:
                es.append(e)
        return x, e, es


class SkipTopResidualNetwork(ResidualNetwork):
    def __init__(self, layers: list[nn.Module], connect_to=1, **kwargs):
        """Skip connections to one fixed early layer (resin.py:178-216).

        Args:
            connect_to: 0 = the input, 1 = output of the first layer, ...
        """
        assert connect_to <= len(layers)
        super().__init__(layers=layers, **kwargs)
        self._residual_layer = connect_to

    def forward_csr(self, gi, x, e):
        es = [e] if self._collect_hidden_edge_embeds else None
        x_res = None
        for i, layer in enumerate(self.layers):
            if i == self._residual_layer:
                x_res = x
            x, e = layer.forward_csr(gi, x, e, relu_in=i > 0, residue=_res(x_res, self._alpha),
                                     alpha_residue=self._alpha)
            if es is not None:
                es.append(e)
        return x, e, es


RESIDUAL_NETWORKS_BY_NAME: dict[str, Any] = {
    "skip1": Skip1ResidualNetwork,
    "skip2": Skip2ResidualNetwork,
    "skip_top": SkipTopResidualNetwork,
}


class ResIN(nn.Module, HyperparametersMixin):
    def __init__(self, *, node_dim: int, edge_dim: int, object_hidden_dim=40,
                 relational_hidden_dim=40, alpha: float = 0.5, n_layers=1,
                 residual_type: str = "skip1", residual_kwargs: dict | None = None):
        """``n_layers`` identical interaction networks with residual connections
        (resin.py:226-295).

        Args:
            node_dim: node feature dimension
            edge_dim: edge feature dimension
            object_hidden_dim: hidden width of the object models
            relational_hidden_dim: hidden width of the relational models
            alpha: strength of the node residual connection
            n_layers: number of interaction networks
            residual_type: 'skip1', 'skip2' or 'skip_top'
            residual_kwargs: extra arguments of the residual network
        """
        super().__init__()
        self.save_hyperparameters()
        residual_kwargs = dict(residual_kwargs or {})
        layers = [
            InteractionNetwork(node_indim=node_dim, edge_indim=edge_dim, node_outdim=node_dim,
                               edge_outdim=edge_dim, node_hidden_dim=object_hidden_dim,
                               edge_hidden_dim=relational_hidden_dim)
            for _ in range(n_layers)
        ]
        if residual_type == "skip2":
            residual_kwargs["node_dim"] = node_dim
            residual_kwargs["edge_dim"] = edge_dim
        self.network = RESIDUAL_NETWORKS_BY_NAME[residual_type](layers, alpha=alpha,
                                                                **residual_kwargs)
        self.node_dim = node_dim
        self.edge_dim = edge_dim
        self._residual_type = residual_type

    @property
    def concat_edge_embeddings_length(self) -> int:
        """Width of the concatenated edge embeddings of all levels (resin.py:282-290)."""
        if self._residual_type == "skip2":
            return self.edge_dim * (len(self.network.layers) // 2 + 1)
        return self.edge_dim * (len(self.network.layers) + 1)

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor):
        return self.network.forward(x, edge_index, edge_attr)

    def forward_csr(self, gi: ops.GraphIndex, x: Tensor, e_csr: Tensor):
        return self.network.forward_csr(gi, x, e_csr)
