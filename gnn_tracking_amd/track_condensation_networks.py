"""Graph track-condensation networks on the fused HIP kernels.

Reference: models/track_condensation_networks.py:118-403 (``ModularGraphTCN``, ``GraphTCN``,
``PreTrainedECGraphTCN``), models/mlp.py:65-178 (``ResFCNN``, ``HeterogeneousResFCNN``) and
models/graph_construction.py:25-132 (the ``GraphConstruction*FCNN`` embedding networks).  Same constructor
keywords, ``hparams``, ``state_dict`` keys and output dict
``{"W", "H", "B", "ec_hit_mask", "ec_edge_mask"}``.

Data flow: edge classifier -> threshold cut on ``W`` (edge compaction,
``Data.edge_subgraph``) -> optional orphan-node masking (``Data.subgraph``: relabelled
``edge_index``) -> node / edge encoders -> track-condenser ``ResIN`` on the pruned graph
-> beta head (clamped sigmoid, fused epilogue) and cluster-coordinate head.  The MLPs and
interaction networks are the fused kernels of this package; the mask / compaction /
relabel bookkeeping between them is two stream compactions (graph_cut.py, csrc/compact.hip)
whose index lists gather the attributes.
"""

from __future__ import annotations

import math
import os

import torch
from torch import Tensor, nn

from . import _capi, edge_order, graph_cut, ops, ops_ml, precision
from .edge_classifier import ECForGraphTCN, PerfectEdgeClassification
from .hparams import HyperparametersMixin, assert_feat_dim, obj_from_or_to_hparams
from .mlp import MLP
from .resin import ResIN


#: the residual-FCNN kernel on / off (off: the fused two- / three-layer MLP kernels where they apply, library
#: GEMMs otherwise - A/B measurements and bisecting)
_RESFCNN_KERNEL = os.environ.get("GNNTRK_RESFCNN", "1") != "0"


class ResFCNN(nn.Module):
    def __init__(self, *, in_dim: int, hidden_dim: int, out_dim: int, depth: int, alpha: float = 0.6,
                 bias: bool = True):
        """Fully connected network with residual connections (models/mlp.py:65-123):
        L2-normalised input -> encoder -> ``depth-1`` residual hidden layers
        ``x = sqrt(a) x + sqrt(1-a) W relu(x)`` -> decoder on ``relu(x)``.

        Up to ``in_dim`` 64, ``hidden_dim`` 128, ``out_dim`` 32 and any depth the whole network - input
        normalisation, encoder, residual layers, decoder, output scale - is ONE forward and ONE backward
        launch (``ops_ml.res_fcnn``: csrc/resfcnn.hip, fp32 MFMA, activations in registers, one layer's
        weights at a time in LDS).  In bf16 storage ``depth == 1`` (the node encoder of ``ModularGraphTCN``)
        stays on the fused bf16 gather-MLP kernels so that its rows stay bf16.  Wider stacks (hidden
        256-512) are GEMM-shaped work on [N, hidden] x [hidden, hidden] and go to the library GEMM
        (hipBLASLt through ``torch.nn.functional.linear``), with the residual mix as torch device ops.
        """
        super().__init__()
        if depth < 1:
            raise ValueError("Depth must be at least 1")
        self._encoder = nn.Linear(in_dim, hidden_dim, bias=bias)
        self._decoder = nn.Linear(hidden_dim, out_dim, bias=bias)
        self._layers = nn.ModuleList([nn.Linear(hidden_dim, hidden_dim, bias=bias) for _ in range(depth - 1)])
        self._reset_layer_parameters(self._encoder, var=1 / in_dim)
        for layer in self._layers:
            self._reset_layer_parameters(layer, var=2 / hidden_dim)
        self._reset_layer_parameters(self._decoder, var=2 / hidden_dim)
        self._alpha = alpha
        # depth 1 = a two-layer MLP; depth 2 with alpha = 0 (the heterogeneous node encoder of
        # ModularGraphTCN) = a three-layer MLP: both are one fused launch where the storage mode's
        # kernels hold the widths (fp32: in <= 48, hidden <= 64; bf16: 16 input chunks, hidden <= 94)
        self._fusable_depth = depth == 1 or (depth == 2 and alpha == 0)
        self._dims = (in_dim, hidden_dim, out_dim)

    @staticmethod
    def _reset_layer_parameters(layer, var: float):
        layer.reset_parameters()
        for p in layer.parameters():
            nn.init.normal_(p.data, mean=0, std=math.sqrt(var))

    def forward(self, x: Tensor, *, epilogue: int = _capi.EPI_NONE, scale: Tensor | None = None, **ignore) -> Tensor:
        """``scale``: one-element parameter multiplied onto the output (the embedding networks'
        ``_latent_normalization``) - inside the kernel where the residual-FCNN kernel runs."""
        _capi.require_device(x)
        bf16 = precision.use_bf16()
        lin = [self._encoder, *self._layers, self._decoder]
        ws, bs = [l.weight for l in lin], [l.bias for l in lin]
        in_dim, hidden, out_dim = self._dims
        if _RESFCNN_KERNEL and not (bf16 and self._fusable_depth) and epilogue in (_capi.EPI_NONE, _capi.EPI_RELU) \
                and ops_ml.res_fcnn_supported(in_dim, hidden, out_dim, len(lin) - 1):
            # ONE launch for the whole network, any depth (gnntrk_resfcnn_forward: L2 normalisation, encoder,
            # residual layers, decoder, output scale / ReLU; fp32, activations in registers)
            relu = epilogue == _capi.EPI_RELU
            if relu and scale is not None:
                # (the kernel applies the scale BEFORE its ReLU, the paths below - and the reference's callers - after:
                #  the two differ for a negative scale; no model of the reference combines them - scale outside)
                return ops_ml.res_fcnn(x.float(), ws, bs, alpha=self._alpha, normalize=True, out_relu=True) * scale
            return ops_ml.res_fcnn(x.float(), ws, bs, alpha=self._alpha, normalize=True, out_relu=relu, scale=scale)
        x = nn.functional.normalize(x.float(), p=2.0, dim=1, eps=1e-12)
        if self._fusable_depth:
            # bf16 storage: one fused MLP launch where the instantiations hold the widths, the same operator as
            # library GEMMs otherwise (ops.fused_mlp decides) - the rows stay bf16 either way, so that what
            # follows keeps running in that mode
            xs = x.to(torch.bfloat16) if bf16 else x
            if bf16 or ops._fused_supported([ops.Seg(xs)], ws, bs, False, epilogue):
                y = ops.fused_mlp([ops.Seg(xs)], ws, bs, epilogue=epilogue)
                return y if scale is None else y.float() * scale
        # wider than the kernels' limits (hidden > 128, in > 64, out > 32): library GEMMs
        x = self._encoder(x)
        for layer in self._layers:
            x = math.sqrt(self._alpha) * x + math.sqrt(1 - self._alpha) * layer(torch.relu(x))
        x = self._decoder(torch.relu(x))
        x = torch.relu(x) if epilogue == _capi.EPI_RELU else x
        return x if scale is None else x * scale


class GraphConstructionFCNN(ResFCNN, HyperparametersMixin):
    def __init__(self, *, in_dim: int, hidden_dim: int, out_dim: int, depth: int, alpha: float = 0.6):
        """Embedding network of the metric-learning graph construction
        (models/graph_construction.py:25-53): ``ResFCNN`` without biases plus a learnable
        normalisation of the latent space; ``forward(data) -> {"H": ...}``."""
        super().__init__(in_dim=in_dim, hidden_dim=hidden_dim, out_dim=out_dim, depth=depth, alpha=alpha,
                         bias=False)
        self._latent_normalization = nn.Parameter(torch.tensor([1.0]), requires_grad=True)
        self.save_hyperparameters()

    def forward(self, data) -> dict[str, Tensor]:
        return {"H": ResFCNN.forward(self, data.x, scale=self._latent_normalization).float()}


def get_pixel_mask(layer: Tensor) -> Tensor:
    """models/mlp.py:123-124: hits on the 18 pixel layers."""
    return torch.isin(layer, torch.arange(18, device=layer.device))


class HeterogeneousResFCNN(nn.Module):
    def __init__(self, *, in_dim: int, out_dim: int, hidden_dim: int, depth: int, alpha: float = 0.6,
                 bias: bool = True):
        """Separate ``ResFCNN`` s for pixel and strip hits (models/mlp.py:127-178).  As in the
        reference the two embeddings are stacked, pixel rows first: the hits are expected
        sorted by pixel, then strip."""
        super().__init__()
        kw = dict(in_dim=in_dim, hidden_dim=hidden_dim, out_dim=out_dim, depth=depth, alpha=alpha, bias=bias)
        self.pixel_fcnn = ResFCNN(**kw)
        self.strip_fcnn = ResFCNN(**kw)

    def forward(self, x: Tensor, layer: Tensor, *, epilogue: int = _capi.EPI_NONE, scale: Tensor | None = None) -> Tensor:
        pixel_mask = get_pixel_mask(layer)
        if "PYTEST_CURRENT_TEST" not in os.environ and (pixel_mask.all() or not pixel_mask.any()):
            raise ValueError("All or no pixel data found; this doesn't make sense with heterogeneous model")
        embed_pixel = self.pixel_fcnn(x[pixel_mask], epilogue=epilogue, scale=scale)
        embed_strip = self.strip_fcnn(x[~pixel_mask], epilogue=epilogue, scale=scale)
        return torch.vstack([embed_pixel, embed_strip])


class GraphConstructionHeteroResFCNN(HeterogeneousResFCNN, HyperparametersMixin):
    def __init__(self, *, in_dim: int, hidden_dim: int, out_dim: int, depth: int, alpha: float = 0.6):
        """Fully heterogeneous embedding network (models/graph_construction.py:56-87)."""
        super().__init__(in_dim=in_dim, hidden_dim=hidden_dim, out_dim=out_dim, depth=depth, alpha=alpha,
                         bias=False)
        self._latent_normalization = nn.Parameter(torch.tensor([1.0]), requires_grad=True)
        self.save_hyperparameters()

    def forward(self, data) -> dict[str, Tensor]:
        out = HeterogeneousResFCNN.forward(self, data.x, layer=data.layer, scale=self._latent_normalization).float()
        return {"H": out}


class GraphConstructionHeteroEncResFCNN(nn.Module, HyperparametersMixin):
    def __init__(self, *, in_dim: int, hidden_dim_enc: int, hidden_dim: int, out_dim: int, depth_enc: int,
                 depth: int, alpha: float = 0.6):
        """Heterogeneous encoding, shared ``ResFCNN`` behind it
        (models/graph_construction.py:90-132)."""
        super().__init__()
        self.encoder = HeterogeneousResFCNN(in_dim=in_dim, hidden_dim=hidden_dim_enc, out_dim=hidden_dim,
                                            depth=depth_enc, alpha=alpha, bias=False)
        self.fcnn = ResFCNN(in_dim=hidden_dim, hidden_dim=hidden_dim, out_dim=out_dim, depth=depth,
                            alpha=alpha, bias=False)
        self._latent_normalization = nn.Parameter(torch.tensor([1.0]), requires_grad=True)
        self.save_hyperparameters()

    def forward(self, data) -> dict[str, Tensor]:
        assert_feat_dim(data.x, self.hparams.in_dim)
        enc = self.encoder(data.x, layer=data.layer, epilogue=_capi.EPI_RELU)
        assert_feat_dim(enc, self.hparams.hidden_dim)
        out = self.fcnn(enc, scale=self._latent_normalization).float()
        assert_feat_dim(out, self.hparams.out_dim)
        return {"H": out}


class GraphConstructionResIN(nn.Module, HyperparametersMixin):
    def __init__(self, *, node_indim: int, edge_indim: int, h_outdim: int = 8, hidden_dim: int = 40,
                 alpha: float = 0.5, n_layers: int = 1, alpha_fcnn: float = 0.5):
        """Refinement of a metric-learning latent space with a residual stack of interaction
        networks (models/graph_construction.py:136-219): encoders to ``hidden_dim``, ``ResIN``
        with node and edge width ``hidden_dim``, decoder to ``h_outdim``, mixed with the first
        ``h_outdim`` input features.  The interaction networks are fused kernels where ``3 * hidden_dim``
        fits their input width: 128 features in fp32 (csrc/mlp_wide.hip above 48) and 128 slots in bf16
        storage (outputs up to 48 wide where the hidden width is 33 .. 47) - the reference's default
        ``hidden_dim=40`` in both; wider stacks run the same operator as library GEMMs (``ops._wide_mlp``)."""
        super().__init__()
        self.save_hyperparameters()
        self._node_encoder = MLP(node_indim, hidden_dim, hidden_dim=hidden_dim, L=2, bias=False)
        self._edge_encoder = MLP(edge_indim, hidden_dim, hidden_dim=hidden_dim, L=2, bias=False)
        self._resin = ResIN(node_dim=hidden_dim, edge_dim=hidden_dim, object_hidden_dim=hidden_dim,
                            relational_hidden_dim=hidden_dim, n_layers=n_layers, alpha=alpha)
        self._decoder = MLP(hidden_dim, h_outdim, hidden_dim=hidden_dim, L=2, bias=False)
        self._latent_normalization = nn.Parameter(torch.tensor([1.0]), requires_grad=True)

    def forward(self, data) -> dict[str, Tensor]:
        x_fcnn = data.x[:, :self.hparams.h_outdim]
        assert_feat_dim(data.x, self.hparams.node_indim)
        assert_feat_dim(data.edge_attr, self.hparams.edge_indim)
        x_in, ea_in = data.x, data.edge_attr
        # The stack runs in the graph index's CSR edge order and its edge embeddings are not part of the
        # output: the four dataset features per edge are brought into that order ONCE (16 bytes per edge)
        # instead of the hidden_dim-wide encoder output going there and the last edge embedding coming back
        # (160 bytes per edge each way at the default width, forward and backward)
        gi = ops.graph_index(data.edge_index, int(x_in.shape[0]))
        if precision.use_bf16():   # bf16 storage: the dataset's fp32 features are converted once
            from . import ops_bf16
            x_in, ea_in = ops_bf16.to_rows16(x_in), ops_bf16.to_rows16(ea_in, gi.perm)
        else:
            ea_in = ops.permute_rows(ea_in.detach() if not ea_in.requires_grad else ea_in, gi.perm, scatter=False)
        x = self._node_encoder(x_in)
        edge_attr = self._edge_encoder(ea_in)
        x, _, _ = self._resin.forward_csr(gi, x, edge_attr)
        assert_feat_dim(x, self.hparams.hidden_dim)
        delta = self._decoder(x).float()
        assert_feat_dim(delta, self.hparams.h_outdim)
        h = self.hparams.alpha_fcnn * x_fcnn + (1 - self.hparams.alpha_fcnn) * delta
        return {"H": h * self._latent_normalization}


class ModularGraphTCN(nn.Module, HyperparametersMixin):
    def __init__(self, *, ec: nn.Module | None = None, hc_in: nn.Module, node_indim: int,
                 edge_indim: int, h_dim: int = 5, e_dim: int = 4, h_outdim: int = 2,
                 hidden_dim: int = 40, feed_edge_weights: bool = False, ec_threshold: float = 0.5,
                 mask_orphan_nodes: bool = False, use_ec_embeddings_for_hc: bool = False,
                 alpha_latent: float = 0.0, n_embedding_coords: int = 0,
                 heterogeneous_node_encoder: bool = False):
        """Track condensation network on pre-constructed graphs: optional edge classifier,
        node / edge encoders to ``(h_dim, e_dim)``, a track condenser ``hc_in`` and the
        beta / cluster heads (arguments as models/track_condensation_networks.py:118-170).
        """
        super().__init__()
        self.save_hyperparameters(ignore=["ec", "hc_in"])
        self.relu = nn.ReLU()
        self.ec = obj_from_or_to_hparams(self, "ec", ec)
        self.hc_in = obj_from_or_to_hparams(self, "hc_in", hc_in)
        node_enc_indim, edge_enc_indim = node_indim, edge_indim
        if use_ec_embeddings_for_hc:
            ec_node_latent_dim, ec_edge_latent_dim = ec.latent_dim
            node_enc_indim += int(ec_node_latent_dim)
            edge_enc_indim += int(ec_edge_latent_dim)
        edge_enc_indim += int(feed_edge_weights)
        self.hc_edge_encoder = MLP(edge_enc_indim, e_dim, hidden_dim=hidden_dim, L=2, bias=False)
        if not heterogeneous_node_encoder:
            self.hc_node_encoder = ResFCNN(in_dim=node_enc_indim, out_dim=h_dim, hidden_dim=hidden_dim,
                                           depth=1, bias=False, alpha=0)
        else:
            self.hc_node_encoder = HeterogeneousResFCNN(in_dim=node_enc_indim, out_dim=h_dim,
                                                        hidden_dim=hidden_dim, depth=2, bias=False, alpha=0)
        self.p_beta = MLP(h_dim, 1, hidden_dim, L=3)
        self.p_cluster = MLP(h_dim, h_outdim, hidden_dim, L=3)
        self._latent_normalization = nn.Parameter(torch.Tensor([1.0]), requires_grad=True)

    def forward(self, data) -> dict[str, Tensor | None]:
        edge_weights_unmasked = edge_mask = hit_mask = None
        if self.ec is not None:
            ec_result = self.ec(data)
            # (the cut below consumes W in edge_index order: plain tensors from here on)
            data.edge_weights = edge_order.as_tensor(ec_result["W"]).reshape((-1, 1))
            data.ec_node_embedding = ec_result.get("node_embedding", None)
            data.ec_edge_embedding = edge_order.as_tensor(ec_result.get("edge_embedding", None))
            edge_weights_unmasked = data.edge_weights.squeeze()
            # threshold cut and orphan masking: device stream compactions (graph_cut.py)
            # (edge_attr of the kept edges is only read by the next encoder: through a fused gather,
            # unless something else below needs the rows themselves)
            lazy = () if (self.hparams.mask_orphan_nodes or self.hparams.use_ec_embeddings_for_hc
                          or self.hparams.feed_edge_weights) else ("edge_attr",)
            # (the cut graph stays inside this function: only what is read below is carried over)
            only = None
            if not self.hparams.mask_orphan_nodes:
                only = {"edge_attr"} | ({"ec_edge_embedding"} if self.hparams.use_ec_embeddings_for_hc else set()) \
                    | ({"edge_weights"} if self.hparams.feed_edge_weights else set())
            data, edge_mask = graph_cut.edge_cut(data, data.edge_weights, self.hparams.ec_threshold, lazy=lazy,
                                                 only=only)
            if self.hparams.mask_orphan_nodes:
                data, hit_mask = graph_cut.drop_orphans(data)
            else:
                hit_mask = torch.ones(data.num_nodes, dtype=torch.bool, device=data.x.device)
                # (lets the condensation losses skip three boolean-mask gathers and their host
                # synchronisations: they would select every hit)
                hit_mask._gnntrk_all_true = True
        if self.ec is None and self.hparams.feed_edge_weights:
            data.edge_weights = data.ec_score.reshape((-1, 1))

        lazy_ea = getattr(data, "_lazy_rows", {}).get("edge_attr") if self.ec is not None else None
        _edge_attrs, _xs = [data.edge_attr if lazy_ea is None else lazy_ea[0]], [data.x]
        if self.hparams.use_ec_embeddings_for_hc:
            assert data.ec_edge_embedding is not None
            assert data.ec_node_embedding is not None
            _edge_attrs.append(data.ec_edge_embedding)
            _xs.append(data.ec_node_embedding)
        if self.hparams.feed_edge_weights:
            _edge_attrs.append(data.edge_weights)
        # (in bf16 storage mode the EC embeddings are bf16: concatenate in one dtype)
        cdt = torch.bfloat16 if precision.use_bf16() else torch.float32
        x = torch.cat([t.to(cdt) for t in _xs], dim=1) if len(_xs) > 1 else _xs[0]
        edge_attrs = (torch.cat([t.to(cdt) for t in _edge_attrs], dim=1) if len(_edge_attrs) > 1
                      else _edge_attrs[0]).to(cdt)
        # relu(encoder(.)) with the ReLU fused as the kernels' epilogue
        h_hc = self.hc_node_encoder(x, layer=getattr(data, "layer", None), epilogue=_capi.EPI_RELU)
        if lazy_ea is None:
            edge_attr_hc = self.hc_edge_encoder.fused([ops.Seg(edge_attrs)], epilogue=_capi.EPI_RELU)
        else:   # rows of the uncut tensor, gathered inside the kernel by the kept-edge index
            edge_attr_hc = self.hc_edge_encoder.fused([ops.Seg(edge_attrs, lazy_ea[1])], epilogue=_capi.EPI_RELU,
                                                      n_rows=int(lazy_ea[1].numel()))

        h_hc, _, _ = self.hc_in(h_hc, data.edge_index, edge_attr_hc)
        # epsilon + (1 - 2 epsilon) * sigmoid(.): the clamped-sigmoid epilogue
        epsilon = 1e-6
        beta = self.p_beta.fused([ops.Seg(h_hc)], epilogue=_capi.EPI_SIGMOID, ca=epsilon,
                                 cb=1 - 2 * epsilon)
        assert not torch.isnan(beta).any()
        h = self.p_cluster(h_hc).float()  # H, B and W leave in fp32 in both storage modes
        if alpha_residue := self.hparams.alpha_latent:
            nec: int = self.hparams.n_embedding_coords
            assert nec > 0
            assert nec <= h.shape[1]
            residual = nn.functional.pad(data.x[:, :nec], (0, h.shape[1] - nec))
            h = math.sqrt(alpha_residue) * residual + math.sqrt(1 - alpha_residue) * h
        h = h * self._latent_normalization
        return {"W": edge_weights_unmasked, "H": h, "B": beta.squeeze(), "ec_hit_mask": hit_mask,
                "ec_edge_mask": edge_mask}


class GraphTCN(nn.Module, HyperparametersMixin):
    def __init__(self, node_indim: int, edge_indim: int, *, h_dim=5, e_dim=4, h_outdim=2,
                 hidden_dim=40, L_ec=3, L_hc=3, alpha_ec: float = 0.5, alpha_hc: float = 0.5,
                 **kwargs):
        """``ModularGraphTCN`` with ``ECForGraphTCN`` as edge classifier and a ``ResIN``
        stack as track condenser (models/track_condensation_networks.py:311-385)."""
        super().__init__()
        self.save_hyperparameters()
        ec = ECForGraphTCN(node_indim=node_indim, edge_indim=edge_indim, hidden_dim=hidden_dim,
                           interaction_node_dim=h_dim, interaction_edge_dim=e_dim, L_ec=L_ec,
                           alpha=alpha_ec)
        hc_in = ResIN(node_dim=h_dim, edge_dim=e_dim, object_hidden_dim=hidden_dim,
                      relational_hidden_dim=hidden_dim, alpha=alpha_hc, n_layers=L_hc)
        self._gtcn = ModularGraphTCN(ec=ec, hc_in=hc_in, node_indim=node_indim,
                                     edge_indim=edge_indim, h_dim=h_dim, e_dim=e_dim,
                                     h_outdim=h_outdim, hidden_dim=hidden_dim, **kwargs)

    def forward(self, data) -> dict[str, Tensor | None]:
        return self._gtcn.forward(data=data)


class PreTrainedECGraphTCN(nn.Module, HyperparametersMixin):
    def __init__(self, ec, *, node_indim: int, edge_indim: int, h_dim=5, e_dim=4, h_outdim=2,
                 hidden_dim=40, L_hc=3, alpha_hc: float = 0.5, **kwargs):
        """``GraphTCN`` with a given (pre-trained) edge classifier
        (models/track_condensation_networks.py:457-517)."""
        super().__init__()
        self.save_hyperparameters(ignore=["ec"])
        ec = obj_from_or_to_hparams(self, "ec", ec)
        hc_in = ResIN(node_dim=h_dim, edge_dim=e_dim, object_hidden_dim=hidden_dim,
                      relational_hidden_dim=hidden_dim, alpha=alpha_hc, n_layers=L_hc)
        self._gtcn = ModularGraphTCN(ec=ec, hc_in=hc_in, node_indim=node_indim,
                                     edge_indim=edge_indim, h_dim=h_dim, e_dim=e_dim,
                                     h_outdim=h_outdim, hidden_dim=hidden_dim, **kwargs)

    def forward(self, data) -> dict[str, Tensor | None]:
        return self._gtcn.forward(data=data)


class PerfectECGraphTCN(nn.Module, HyperparametersMixin):
    def __init__(self, *, node_indim: int, edge_indim: int, h_dim=5, e_dim=4, h_outdim=2, hidden_dim=40,
                 L_hc=3, alpha_hc: float = 0.5, ec_tpr=1.0, ec_tnr=1.0, **kwargs):
        """``GraphTCN`` with the truth-based edge classifier
        (models/track_condensation_networks.py:389-454)."""
        super().__init__()
        self.save_hyperparameters()
        ec = PerfectEdgeClassification(tpr=ec_tpr, tnr=ec_tnr)
        hc_in = ResIN(node_dim=h_dim, edge_dim=e_dim, object_hidden_dim=hidden_dim,
                      relational_hidden_dim=hidden_dim, alpha=alpha_hc, n_layers=L_hc)
        self._gtcn = ModularGraphTCN(ec=ec, hc_in=hc_in, node_indim=node_indim, edge_indim=edge_indim,
                                     h_dim=h_dim, e_dim=e_dim, h_outdim=h_outdim, hidden_dim=hidden_dim,
                                     **kwargs)

    def forward(self, data) -> dict[str, Tensor | None]:
        return self._gtcn.forward(data=data)


class GraphTCNForMLGCPipeline(nn.Module, HyperparametersMixin):
    def __init__(self, *, node_indim: int, edge_indim: int, h_dim=5, e_dim=4, h_outdim=2, hidden_dim=40,
                 L_hc=3, alpha_hc: float = 0.5, **kwargs):
        """``GraphTCN`` without an edge classifier, behind a metric-learning graph construction
        (models/track_condensation_networks.py:522-582)."""
        super().__init__()
        self.save_hyperparameters(ignore=["ec"])
        hc_in = ResIN(node_dim=h_dim, edge_dim=e_dim, object_hidden_dim=hidden_dim,
                      relational_hidden_dim=hidden_dim, alpha=alpha_hc, n_layers=L_hc)
        self._gtcn = ModularGraphTCN(hc_in=hc_in, node_indim=node_indim, edge_indim=edge_indim, h_dim=h_dim,
                                     e_dim=e_dim, h_outdim=h_outdim, hidden_dim=hidden_dim, **kwargs)

    def forward(self, data) -> dict[str, Tensor | None]:
        return self._gtcn.forward(data=data)
