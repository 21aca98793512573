"""Interaction network layer on the fused HIP kernels.

Reference: models/interaction_network.py:12-103 (a PyG ``MessagePassing`` with
``aggr="add"``, ``flow="source_to_target"``).  Same constructor keywords,
``hparams``, ``state_dict`` keys (``relational_model.layers.*``,
``object_model.layers.*``) and return value ``(x_tilde, e_tilde)``.

Data flow per call (j = edge_index[0] source, i = edge_index[1] target):

    e~[k]   = relational( [x[i_k], x[j_k], e[k]] )      one fused gather-MLP kernel
    aggr[n] = sum_{k: i_k = n} e~[k]                    deterministic CSR segment sum
    x~[n]   = object( [x[n], aggr[n]] )                 one fused MLP kernel

Edges are processed in target-sorted (CSR) order; ``forward`` converts from/to the
caller's COO order, ``forward_csr`` is the no-conversion entry the residual stacks use.
"""

from __future__ import annotations

import torch
from torch import Tensor, nn

from . import _capi, ops, ops_bf16
from .hparams import HyperparametersMixin, assert_feat_dim
from .mlp import MLP


class InteractionNetwork(nn.Module, HyperparametersMixin):
    def __init__(self, *, node_indim: int, edge_indim: int, node_outdim=3, edge_outdim=4,
                 node_hidden_dim=40, edge_hidden_dim=40, aggr="add"):
        """Relational model (edge MLP) + object model (node MLP).

        Args:
            node_indim: node feature dimension
            edge_indim: edge feature dimension
            node_outdim: output node feature dimension
            edge_outdim: output edge feature dimension
            node_hidden_dim: hidden width of the object model
            edge_hidden_dim: hidden width of the relational model
            aggr: message aggregation; only "add" (the reference's default and only
                use) is implemented
        """
        super().__init__()
        self.save_hyperparameters()
        if aggr != "add":
            raise NotImplementedError("only aggr='add' is implemented")
        self.relational_model = MLP(2 * node_indim + edge_indim, edge_outdim, edge_hidden_dim)
        self.object_model = MLP(node_indim + edge_outdim, node_outdim, node_hidden_dim)

    def forward_csr(self, gi: ops.GraphIndex, x: Tensor, e_csr: Tensor, *, relu_in: bool = False,
                    residue: Tensor | None = None, alpha_residue: float = 0.0):
        """``x`` [N,Dn]; ``e_csr`` [E,De] in CSR order.  ``relu_in`` fuses the ReLU the
        residual stacks apply to both inputs (resin.py:103-104).  With ``residue`` the
        node output is ``sqrt(a)*residue + sqrt(1-a)*x~`` (resin.py:26), fused as the
        object model's epilogue.  Returns ``(x_out, e_tilde_csr)``."""
        rel_segs = [
            ops.Seg(x, gi.tgt, relu_in, ("tgt", gi)),
            ops.Seg(x, gi.src, relu_in, ("src", gi)),
            ops.Seg(e_csr, None, relu_in),
        ]
        lin = self.relational_model.linears()
        if x.dtype == torch.bfloat16 and ops._fused_supported(rel_segs, [m.weight for m in lin],
                                                              [m.bias for m in lin], True):
            # bf16 storage: relational model + aggregation as one autograd node, so that the
            # backward kernel takes "direct" and "through aggr" gradients as two terms
            e_tilde, aggr = ops_bf16.in_edge(rel_segs, [m.weight for m in lin], [m.bias for m in lin],
                                             gi, gi.n_edges)
        elif (x.dtype == torch.float32 and e_csr.dtype == torch.float32 and gi.n_edges > 0
              and not ops._fused_supported(rel_segs, [m.weight for m in lin], [m.bias for m in lin], False)
              and ops._wide_kernel_supported(rel_segs, [m.weight for m in lin], None)):
            # fp32 at widths of the wide kernels: the same single node (the aggregation's gradient reaches the
            # backward kernel as a gathered term instead of an edge-sized tensor)
            e_tilde, aggr = ops.in_edge_wide(rel_segs, [m.weight for m in lin], [m.bias for m in lin],
                                             gi, gi.n_edges)
        else:
            e_tilde = self.relational_model.fused(rel_segs, n_rows=gi.n_edges)
            aggr = ops.segment_sum(e_tilde, gi, "tgt")
        x_obj, aggr_obj = ops_bf16.node_tap(x, aggr) if x.dtype == torch.bfloat16 else (x, aggr)
        segs = [ops.Seg(x_obj, None, relu_in), ops.Seg(aggr_obj)]
        if residue is not None:
            ca, cb = float(alpha_residue) ** 0.5, (1.0 - float(alpha_residue)) ** 0.5
            x_out = self.object_model.fused(segs, epilogue=_capi.EPI_RESIDUAL, ca=ca, cb=cb,
                                            res=residue)
        else:
            x_out = self.object_model.fused(segs)
        return x_out, e_tilde

    def forward(self, x: Tensor, edge_index: Tensor, edge_attr: Tensor) -> tuple[Tensor, Tensor]:
        """Returns ``(output node embedding, output edge embedding)``; the edge
        embedding is in the order of ``edge_index``."""
        assert_feat_dim(x, self.hparams.node_indim)
        assert_feat_dim(edge_attr, self.hparams.edge_indim)
        gi = ops.graph_index(edge_index, x.shape[0])
        e_csr = ops.permute_rows(edge_attr, gi.perm, scatter=False)
        x_tilde, e_tilde = self.forward_csr(gi, x, e_csr)
        return x_tilde, ops.permute_rows(e_tilde, gi.perm, scatter=True)
