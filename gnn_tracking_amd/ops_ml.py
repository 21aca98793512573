"""Autograd wiring of the metric-learning stage's kernels.

* ``res_fcnn``: the residual fully connected network (reference ``models/mlp.py:65-120``) as ONE forward
  and ONE backward launch (``gnntrk_resfcnn_forward`` / ``_backward``, csrc/resfcnn.hip);
* ``hinge_terms``: the two edge-list reductions of ``GraphConstructionHingeEmbeddingLoss``
  (``metrics/losses/metric_learning.py:14-55``) with the reference's edge selection applied inside the
  kernels (``gnntrk_hinge_forward`` / ``_backward``, csrc/hinge.hip).

No CPU path: the tensors must live on the GPU (``_capi.require_device``).
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
from torch import Tensor

from . import _capi, ops
from .ops import _p, _stream, _ws


def res_fcnn_supported(in_dim: int, hidden: int, out_dim: int, depth: int) -> bool:
    """Shapes ``gnntrk_resfcnn_*`` instantiates (include/gnntrk.h)."""
    return (1 <= in_dim <= _capi.RESFCNN_MAX_IN and 1 <= hidden <= _capi.RESFCNN_MAX_WIDTH
            and 1 <= out_dim <= _capi.RESFCNN_MAX_OUT and 1 <= depth <= _capi.RESFCNN_MAX_HIDDEN + 1)


def _model(x: Tensor, weights: Sequence[Tensor], biases: Sequence[Optional[Tensor]], alpha: float, normalize: bool,
           out_relu: bool, scale: Optional[Tensor]) -> _capi.ResFcnn:
    m = _capi.ResFcnn()
    n_hidden = len(weights) - 2
    m.W_enc, m.b_enc = _p(weights[0]), _p(biases[0])
    for i in range(n_hidden):
        m.W_hid[i], m.b_hid[i] = _p(weights[1 + i]), _p(biases[1 + i])
    m.W_dec, m.b_dec = _p(weights[-1]), _p(biases[-1])
    m.out_scale = _p(scale)
    m.in_dim, m.hidden, m.out_dim, m.n_hidden = weights[0].shape[1], weights[0].shape[0], weights[-1].shape[0], n_hidden
    m.alpha, m.normalize, m.out_relu = float(alpha), int(normalize), int(out_relu)
    return m


class _ResFCNN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, alpha: float, normalize: bool, out_relu: bool, n_layers: int, grad_on: bool, *params):
        weights = [w.contiguous() for w in params[:n_layers]]
        biases = [None if b is None else b.contiguous() for b in params[n_layers:]]
        _capi.require_device(x, *weights)
        lib = _capi.load()
        x = x.contiguous()
        if x.dtype != torch.float32 or any(w.dtype != torch.float32 for w in weights):
            raise TypeError("res_fcnn: fp32 rows and parameters expected")
        n = int(x.shape[0])
        if x.shape[1] != weights[0].shape[1]:
            raise AssertionError(f"Expected feature dimension {weights[0].shape[1]}, got {x.shape[1]}")
        sc = None if scale is None else scale.detach().to(torch.float32).contiguous()
        m = _model(x, weights, biases, alpha, normalize, out_relu, sc)
        out = torch.empty(n, m.out_dim, dtype=torch.float32, device=x.device)
        # (needs_input_grad is True under no_grad() too while the parameters are trainable, and inside forward()
        #  grad mode is always off: the caller's grad mode comes in as an argument - inference must not pay for
        #  the saved activations)
        need = grad_on and any(ctx.needs_input_grad)
        acts = None
        if need and n > 0:
            hp = int(lib.gnntrk_resfcnn_hidden_pad(m.hidden))
            acts = torch.empty(m.n_hidden + 1, n, hp, dtype=torch.float32, device=x.device)
        ws = _ws(lib.gnntrk_resfcnn_forward_workspace_bytes(C.byref(m)), x)
        _capi.check(lib.gnntrk_resfcnn_forward(C.byref(m), _p(x), int(x.stride(0)) if n > 1 else int(x.shape[1]), n,
                                               _p(out), m.out_dim, _p(acts), _p(ws), ws.numel(), _stream(x)), lib)
        ctx.cfg = (float(alpha), bool(normalize), bool(out_relu), n_layers)
        ctx.save_for_backward(x, sc, out if out_relu else None, acts, *weights, *[b for b in biases if b is not None])
        ctx.has_bias = [b is not None for b in biases]
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _capi.load()
        alpha, normalize, out_relu, nl = ctx.cfg
        x, sc, out, acts = ctx.saved_tensors[:4]
        weights = list(ctx.saved_tensors[4:4 + nl])
        rest = list(ctx.saved_tensors[4 + nl:])
        biases = [rest.pop(0) if hb else None for hb in ctx.has_bias]
        n = int(x.shape[0])
        m = _model(x, weights, biases, alpha, normalize, out_relu, sc)
        g = g.contiguous().to(torch.float32)
        gW = [torch.empty_like(w) for w in weights]
        gb = [None if b is None else torch.empty_like(b) for b in biases]
        gs = None if sc is None else torch.empty_like(sc)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gr = _capi.ResFcnnGrads()
        gr.W_enc, gr.b_enc = _p(gW[0]), _p(gb[0])
        for i in range(nl - 2):
            gr.W_hid[i], gr.b_hid[i] = _p(gW[1 + i]), _p(gb[1 + i])
        gr.W_dec, gr.b_dec, gr.out_scale = _p(gW[-1]), _p(gb[-1]), _p(gs)
        ws = _ws(lib.gnntrk_resfcnn_backward_workspace_bytes(C.byref(m), n), x)
        _capi.check(lib.gnntrk_resfcnn_backward(C.byref(m), _p(x), int(x.stride(0)) if n > 1 else int(x.shape[1]), n,
                                                _p(acts), _p(out), m.out_dim, _p(g), int(g.stride(0)) if n > 1 else m.out_dim,
                                                _p(gx), int(x.shape[1]), C.byref(gr), 0, _p(ws), ws.numel(), _stream(x)), lib)
        return (gx, gs, None, None, None, None, None, *gW, *gb)


def res_fcnn(x: Tensor, weights: Sequence[Tensor], biases: Sequence[Optional[Tensor]], *, alpha: float,
             normalize: bool = True, out_relu: bool = False, scale: Optional[Tensor] = None) -> Tensor:
    """``weights = [W_enc, W_hidden_1 .., W_dec]`` (``nn.Linear`` storage), ``biases`` likewise (entries may be
    None).  ``scale``: one-element parameter multiplied onto the output (``_latent_normalization``)."""
    return _ResFCNN.apply(x, scale, float(alpha), bool(normalize), bool(out_relu), len(weights), torch.is_grad_enabled(),
                          *weights, *biases)


# ----------------------------------------------------------------------------------- hinge loss
def _hinge_args(x: Tensor, mask: Optional[Tensor], pid: Optional[Tensor], r_emb: float, p: float, repulsive: bool):
    a = _capi.HingeArgs()
    a.x, a.dim, a.x_stride, a.n_nodes = _p(x), int(x.shape[1]), int(x.stride(0)) if x.shape[0] > 1 else int(x.shape[1]), int(x.shape[0])
    a.node_mask, a.particle_id = _p(mask), _p(pid)
    a.r_emb, a.p, a.repulsive = float(r_emb), float(p), int(repulsive)
    return a


class _HingeTerm(torch.autograd.Function):
    """``(loss, count, denom)`` of one edge list: ``loss = sum over the selected edges / denom``."""

    @staticmethod
    def forward(ctx, x, edges, mask, pid, norm, r_emb: float, p: float, repulsive: bool):
        _capi.require_device(x, edges)
        lib = _capi.load()
        x = x.contiguous()
        if x.dtype != torch.float32:
            raise TypeError("hinge loss: fp32 embedding expected")
        if edges.dim() != 2 or edges.shape[0] != 2 or edges.dtype != torch.int64:
            raise ValueError("hinge loss: edges must be int64 [2, E]")
        e = edges if edges.stride(1) == 1 or edges.shape[1] <= 1 else edges.contiguous()
        n_e = int(e.shape[1])
        if ops._VALIDATE and n_e > 0:
            # (GNNTRK_VALIDATE, as for ops.graph_index: ids outside [0, N) raise as torch's indexing would - a host
            #  synchronisation, so off by default; unchecked ids out of range read out of bounds)
            lo, hi = int(e.min()), int(e.max())
            if lo < 0 or hi >= int(x.shape[0]):
                raise IndexError(f"hinge loss: node ids in [{lo}, {hi}] outside [0, {int(x.shape[0])})")
        # the kernel reads one byte per node of the mask and one int64 per node of the particle ids: any other
        # dtype is converted here (the reference's boolean / integer indexing takes them all), sizes are checked
        n_nodes = int(x.shape[0])
        m8 = None
        if mask is not None:
            m8 = (mask.view(torch.uint8) if mask.dtype in (torch.bool, torch.uint8) else (mask != 0).view(torch.uint8)).contiguous()
            if m8.numel() != n_nodes or m8.device != x.device:
                raise ValueError(f"hinge loss: node_mask must hold one value per node ({n_nodes}) on {x.device}")
        if pid is not None:
            if pid.is_floating_point() or pid.dtype == torch.bool:
                raise TypeError("hinge loss: particle_id must be an integer tensor")
            pid = pid.to(torch.int64).contiguous()
            if pid.numel() != n_nodes or pid.device != x.device:
                raise ValueError(f"hinge loss: particle_id must hold one value per node ({n_nodes}) on {x.device}")
        nrm = None if norm is None else norm.detach().to(torch.float32).reshape(1).contiguous()
        out = torch.empty(3, dtype=torch.float32, device=x.device)
        a = _hinge_args(x, m8, pid, r_emb, p, repulsive)
        ws = _ws(lib.gnntrk_hinge_workspace_bytes(n_e), x)
        _capi.check(lib.gnntrk_hinge_forward(C.byref(a), _p(e), n_e, int(e.stride(0)) if n_e > 0 else 0, _p(nrm), _p(out),
                                             _p(ws), ws.numel(), _stream(x)), lib)
        ctx.cfg = (float(r_emb), float(p), bool(repulsive))
        ctx.edges = edges
        ctx.save_for_backward(x, m8, pid, out)
        loss, cnt, den = out[0], out[1], out[2]
        ctx.mark_non_differentiable(cnt, den)
        return loss, cnt, den

    @staticmethod
    def backward(ctx, g, _gc, _gd):
        lib = _capi.load()
        x, m8, pid, out = ctx.saved_tensors
        r_emb, p, repulsive = ctx.cfg
        edges = ctx.edges
        gx = torch.zeros_like(x)
        if edges.shape[1] > 0:
            # (cached per edge-list object: the attractive edges of a batch are indexed once)
            gi = ops.graph_index(edges, int(x.shape[0]), validate=False)
            d = _capi.GraphIndex(gi.n_nodes, gi.n_edges, _p(gi.perm), _p(gi.tgt), _p(gi.src), _p(gi.rowptr_t),
                                 _p(gi.rowptr_s), _p(gi.spos), _p(gi.spos_inv))
            a = _hinge_args(x, m8, pid, r_emb, p, repulsive)
            gg = g.detach().to(torch.float32).reshape(1).contiguous()
            _capi.check(lib.gnntrk_hinge_backward(C.byref(a), C.byref(d), _p(gg), _p(out[2:]), _p(gx), int(gx.stride(0)) if x.shape[0] > 1 else int(x.shape[1]),
                                                  0, _stream(x)), lib)
        return gx, None, None, None, None, None, None, None


def hinge_terms(x: Tensor, edges: Tensor, *, node_mask: Optional[Tensor] = None, particle_id: Optional[Tensor] = None,
                norm: Optional[Tensor] = None, r_emb: float = 1.0, p: float = 1.0, repulsive: bool = False):
    """``(loss, n_selected, denom)``: the attractive (``sum d^p``) or repulsive (``sum relu(r_emb - d^p)``) term
    of the hinge loss over ``edges`` ([2, E] int64), restricted to edges with ``node_mask[edges[0]]`` (if given)
    and different particle ids at the two ends (if ``particle_id`` is given), divided by
    ``(norm if given else n_selected) + 1e-9``."""
    return _HingeTerm.apply(x, edges, node_mask, particle_id, norm, float(r_emb), float(p), bool(repulsive))
