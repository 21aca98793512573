"""Object-condensation losses on fused HIP reductions.

Reference: metrics/losses/__init__.py:13-41 (``MultiLossFctReturn``) and
metrics/losses/oc.py:164-436 (``CondensationLossRG``, ``CondensationLossTiger``): same
constructor keywords, ``hparams`` and return type.  Neither the N x K matrices of the Tiger
variant nor the radius graph of the RG variant are materialised: both are sums over
(hit, condensation point) pairs evaluated by ``gnntrk_oc_forward/backward`` (csrc/oc.hip).

``sample_pids < 1`` and Tiger's ``max_n_rep > 0`` (random sub-sampling switches the reference
has for memory) are honoured for parity of the training dynamics; the random draws differ from
the reference's (as they do between its own CPU and GPU runs).
"""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Any

import torch
from torch import Tensor as T

from . import _capi, ops
from .graph_masks import get_good_node_mask_tensors
from .hparams import HyperparametersMixin


@dataclass(kw_only=True)
class MultiLossFctReturn:
    """Return type of loss functions with several terms (losses/__init__.py:13-35)."""

    #: individual loss terms
    loss_dct: dict[str, T]
    #: their weights
    weight_dct: dict[str, T] | dict[str, float]
    #: other things to log
    extra_metrics: dict[str, Any] = field(default_factory=dict)

    def __post_init__(self) -> None:
        assert self.loss_dct.keys() == self.weight_dct.keys()

    @property
    def loss(self) -> T:
        loss = sum(self.weighted_losses.values())
        assert isinstance(loss, torch.Tensor)
        return loss

    @property
    def weighted_losses(self) -> dict[str, T]:
        return {k: v * self.weight_dct[k] for k, v in self.loss_dct.items()}


class MultiLossFct(torch.nn.Module):
    """Base class of loss functions returning ``MultiLossFctReturn``."""

    def forward(self, *args: Any, **kwargs: Any) -> MultiLossFctReturn: ...


#: "auto": the spatial passes from SPATIAL_MIN_HITS hits on (below, sorting the hits costs more than
#: the dense N x K walk); "on" / "off": always / never (tests, measurements)
SPATIAL = os.environ.get("GNNTRK_OC_SPATIAL", "auto")
SPATIAL_MIN_HITS = 16384


def _use_spatial(n: int) -> bool:
    if SPATIAL not in ("auto", "on", "off"):
        raise ValueError(f"losses_oc.SPATIAL must be 'auto', 'on' or 'off', got {SPATIAL!r}")
    return SPATIAL == "on" or (SPATIAL == "auto" and n >= SPATIAL_MIN_HITS)


class _CondensationPotentials(torch.autograd.Function):
    """(attractive, repulsive, coward, noise) and their gradients w.r.t. (beta, x)."""

    @staticmethod
    def forward(ctx, beta, x, particle_id, mask, q_min: float, radius: float, eps_sqrt: float,
                mode: int, keep: float = 1.0, seed: int = 0, cap_nbr=None, stash=None):
        """``stash`` (a dict): the condensation-point selection of a call is left in it / taken from
        it, so that a second pass over the same hits (``max_n_rep`` sub-sampling) does not select -
        sort and scan - again."""
        _capi.require_device(beta, x, particle_id, mask)
        lib = _capi.load()
        dev = x.device
        beta_c = beta.detach().to(torch.float32).contiguous()
        x_c = x.detach().to(torch.float32).contiguous()
        pid = particle_id.to(torch.int64).contiguous()
        mask8 = mask.to(torch.uint8).contiguous()
        n, dim = int(x_c.shape[0]), int(x_c.shape[1])
        st = ops._stream(x_c)
        if stash is not None and "sel" in stash:
            alphas, gid, n_cp = stash["sel"]
        else:
            alphas = torch.empty(n, dtype=torch.int32, device=dev)
            gid = torch.empty(n, dtype=torch.int32, device=dev)
            n_cp = torch.zeros(1, dtype=torch.int32, device=dev)
            ws = ops._ws(lib.gnntrk_oc_select_workspace_bytes(n), x_c)
            _capi.check(lib.gnntrk_oc_select_cps(ops._p(beta_c), ops._p(pid), ops._p(mask8), n, mode,
                                                 ops._p(alphas), ops._p(gid), ops._p(n_cp), ops._p(ws),
                                                 ws.numel(), st), lib)
            if stash is not None:
                stash["sel"] = (alphas, gid, n_cp)
        a = _capi.OcArgs(ops._p(x_c), ops._p(beta_c), ops._p(pid), ops._p(mask8), ops._p(gid),
                         ops._p(alphas), ops._p(n_cp), n, dim, dim, q_min, radius, eps_sqrt, mode,
                         float(keep), 0, int(seed), ops._p(cap_nbr))
        out = torch.empty(9, dtype=torch.float32, device=dev)
        # large events: the pair loops only visit (chunk of hits, condensation point) pairs within the
        # radius (csrc/oc.hip "spatial" passes; same pairs, same per-pair arithmetic); the buffer the
        # forward fills is what the backward reads
        nb = int(lib.gnntrk_oc_spatial_workspace_bytes(n, dim)) if _use_spatial(n) else 0
        ctx.spatial = None
        if nb:
            ctx.spatial = ops._ws(nb, x_c)
            _capi.check(lib.gnntrk_oc_forward_spatial(C.byref(a), ops._p(out), ops._p(ctx.spatial), nb, st), lib)
        else:
            ws2 = ops._ws(lib.gnntrk_oc_forward_workspace_bytes(n), x_c)
            _capi.check(lib.gnntrk_oc_forward(C.byref(a), ops._p(out), ops._p(ws2), ws2.numel(), st), lib)
        ctx.save_for_backward(beta_c, x_c, pid, mask8, gid, alphas, n_cp, out)
        ctx.cfg = (q_min, radius, eps_sqrt, mode, beta.dtype, x.dtype, float(keep), int(seed))
        ctx.cap_nbr = cap_nbr
        n_rep = out[7].clone()   # a plain count: no gradient, no hold on this node's saved tensors
        ctx.mark_non_differentiable(n_rep)
        return out[0], out[1], out[2], out[3], n_rep

    @staticmethod
    def backward(ctx, g_att, g_rep, g_cow, g_noise, _g_nrep):
        lib = _capi.load()
        beta_c, x_c, pid, mask8, gid, alphas, n_cp, out = ctx.saved_tensors
        q_min, radius, eps_sqrt, mode, bdt, xdt, keep, seed = ctx.cfg
        n, dim = int(x_c.shape[0]), int(x_c.shape[1])
        g = torch.stack([t.to(torch.float32).reshape(()) if t is not None
                         else torch.zeros((), device=x_c.device)
                         for t in (g_att, g_rep, g_cow, g_noise)]).contiguous()
        a = _capi.OcArgs(ops._p(x_c), ops._p(beta_c), ops._p(pid), ops._p(mask8), ops._p(gid),
                         ops._p(alphas), ops._p(n_cp), n, dim, dim, q_min, radius, eps_sqrt, mode,
                         keep, 0, seed, ops._p(ctx.cap_nbr))
        gx = torch.empty_like(x_c)
        gbeta = torch.empty_like(beta_c)
        if ctx.spatial is not None:
            _capi.check(lib.gnntrk_oc_backward_spatial(C.byref(a), ops._p(g), ops._p(out), ops._p(gx), ops._p(gbeta),
                                                       n, ops._p(ctx.spatial), ctx.spatial.numel(),
                                                       ops._stream(x_c)), lib)
        else:
            ws = ops._ws(lib.gnntrk_oc_backward_workspace_bytes(n, dim), x_c)
            _capi.check(lib.gnntrk_oc_backward(C.byref(a), ops._p(g), ops._p(out), ops._p(gx),
                                               ops._p(gbeta), n, ops._p(ws), ws.numel(), ops._stream(x_c)), lib)
        return gbeta.to(bdt), gx.to(xdt), None, None, None, None, None, None, None, None, None, None


class _CondensationLoss(MultiLossFct, HyperparametersMixin):
    _mode = 0
    _eps_sqrt = 1e-9

    def _neighbor_cap(self, x):   # (only the radius-graph variant has one)
        return None

    def _forward(self, *, beta: T, x: T, particle_id: T, reconstructable: T, pt: T,
                 ec_hit_mask: T | None, eta: T, mask_eta: bool) -> MultiLossFctReturn:
        if ec_hit_mask is not None and not getattr(ec_hit_mask, "_gnntrk_all_true", False):
            # model outputs already carry the post-EC node mask, data attributes do not
            particle_id = particle_id[ec_hit_mask]
            reconstructable = reconstructable[ec_hit_mask]
            pt = pt[ec_hit_mask]
            if mask_eta:
                eta = eta[ec_hit_mask]
        mask = get_good_node_mask_tensors(pt=pt, particle_id=particle_id,
                                          reconstructable=reconstructable, eta=eta,
                                          pt_thld=self.hparams.pt_thld,
                                          max_eta=self.hparams.max_eta)
        if self.hparams.sample_pids < 1:
            # oc.py:222-226 / :403-407: a random subset of the hits of interest (the reference does
            # it to save memory; kept for parity of the training dynamics)
            mask = mask & (torch.rand_like(beta, dtype=torch.float16) < self.hparams.sample_pids)
        # If there are no hits left after masking, then we get a NaN loss.
        assert bool(mask.any()), "No hits left after masking"
        args = (beta, x, particle_id, mask, float(self.hparams.q_min), 1.0, self._eps_sqrt, self._mode)
        cap = self._neighbor_cap(x)
        max_n_rep = int(getattr(self.hparams, "max_n_rep", 0) or 0)
        stash = {} if max_n_rep > 0 else None
        att, rep, cow, noise, n_rep = _CondensationPotentials.apply(*args, 1.0, 0, cap, stash)
        if max_n_rep > 0 and int(n_rep) > max_n_rep:
            # oc.py:322-328: keep repulsive pairs with probability max_n_rep / n_rep and scale the
            # normalisation accordingly.  The pairs are chosen by a hash of (seed, hit, condensation
            # point) inside the kernels (no N x K random matrix); the seed comes from torch's
            # generator, so runs are reproducible under torch.manual_seed.  n_rep is reported
            # before the sub-sampling, as the reference does.
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            att, rep, cow, noise, _ = _CondensationPotentials.apply(*args, max_n_rep / float(n_rep), seed, cap, stash)
        losses = {"attractive": att, "repulsive": rep, "coward": cow, "noise": noise}
        weights = {"attractive": 1.0, "repulsive": self.hparams.lw_repulsive,
                   "noise": self.hparams.lw_noise, "coward": self.hparams.lw_coward}
        extra = {"n_rep": n_rep} if self._mode == 1 else {}
        return MultiLossFctReturn(loss_dct=losses, weight_dct=weights, extra_metrics=extra)


class CondensationLossRG(_CondensationLoss):
    _mode = 0
    _eps_sqrt = 1e-9

    def __init__(self, *, lw_repulsive: float = 1.0, lw_noise: float = 0.0, lw_coward: float = 0.0,
                 q_min: float = 0.01, pt_thld: float = 0.9, max_eta: float = 4.0,
                 max_num_neighbors: int = 256, sample_pids: float = 1.0):
        """Condensation loss, radius-graph formulation (oc.py:164-248).

        Args:
            lw_repulsive: weight of the repulsive potential
            lw_noise: weight of the noise loss
            lw_coward: weight of the coward loss
            q_min: minimal charge (object condensation paper)
            pt_thld: pt threshold of the particles of interest
            max_eta: eta threshold of the particles of interest
            max_num_neighbors: neighbour cap of the reference's radius graph (oc.py:115-117).  Applied when
                it can bind (``neighbor_cap``, below): the fused kernel sums over ALL hits within the unit
                radius, which is the reference's result whenever no hit has more neighbours than the cap
            sample_pids: fraction of the hits of interest that take part (random, per call)
        """
        super().__init__()
        self.save_hyperparameters()

    _cap_notice_given = False
    #: "auto" (default): a count pass over the hits (``ops.max_radius_count``) decides - if no hit has more than
    #: ``max_num_neighbors`` other hits inside the unit radius the cap cannot bind and every hit inside the radius of
    #: a condensation point contributes (= the reference); if one has, the cap is applied nearest first (and a notice
    #: is logged once).  The count is a hits x hits pass (4.6 ms at 200 k hits - the fused loss itself only ever looks
    #: at (hit, condensation point) pairs, 0.6 ms), so it runs on the first call and then on every
    #: ``cap_check_every``-th one; calls in between reuse the last decision (an embedding space drifts over many
    #: steps; ``cap_check_every = 1`` decides on every call).  "nearest": always applied nearest first - a
    #: condensation point only repels a hit if it is among that hit's ``max_num_neighbors`` nearest hits (one extra
    #: kNN search per call; what the CPU oracle's radius graph does; torch_cluster itself keeps an
    #: implementation-defined subset: the first ones found, differently on CPU and GPU).  "off": never applied.
    #: Class attributes so that YAML configurations stay those of the reference.
    neighbor_cap = "auto"
    cap_check_every = 64

    def _neighbor_cap(self, x):
        mode = self.neighbor_cap
        if mode == "off":
            return None
        if mode not in ("auto", "nearest"):
            raise ValueError(f"CondensationLossRG.neighbor_cap must be 'auto', 'nearest' or 'off', got {mode!r}")
        k = int(self.hparams.max_num_neighbors)
        if mode == "auto":
            if x.shape[0] - 1 <= k:
                return None
            calls = self._cap_calls = getattr(self, "_cap_calls", -1) + 1
            known = getattr(self, "_cap_binds", None)
            if x.is_cuda and torch.cuda.is_current_stream_capturing():
                # the decision needs a device read: a captured step replays the one its eager warm-up made
                if known is None:
                    raise RuntimeError("CondensationLossRG(neighbor_cap='auto') inside a stream capture needs one eager "
                                       "call first (or neighbor_cap = 'off' / 'nearest')")
                binds = known
            elif known is None or calls % max(int(self.cap_check_every), 1) == 0:
                # (radius widened by 1e-6: the count's fp64 `d2 <= r^2` then covers the fp32 `dist < 1` of the graph)
                binds = self._cap_binds = bool(int(ops.max_radius_count(x, 1.0 + 1e-6)) > k)
            else:
                binds = known
            if not binds:
                return None
            if not CondensationLossRG._cap_notice_given:
                CondensationLossRG._cap_notice_given = True
                import logging
                logging.getLogger("gnn_tracking_amd").warning(
                    "CondensationLossRG: a hit has more than max_num_neighbors=%s hits inside the unit radius - the cap "
                    "is applied nearest first (torch_cluster.radius_graph keeps an implementation-defined subset there, "
                    "differently on CPU and GPU).", k)
        return ops.knn_kth_neighbor(x, k, 1.0)

    def forward(self, *, beta: T, x: T, particle_id: T, reconstructable: T, pt: T,
                ec_hit_mask: T | None = None, eta: T, **kwargs) -> MultiLossFctReturn:
        # NB: like the reference (oc.py:207-213) eta is NOT sliced by ec_hit_mask here
        return self._forward(beta=beta, x=x, particle_id=particle_id,
                             reconstructable=reconstructable, pt=pt, ec_hit_mask=ec_hit_mask,
                             eta=eta, mask_eta=False)


class CondensationLossTiger(_CondensationLoss):
    _mode = 1
    _eps_sqrt = 0.0

    def __init__(self, *, lw_repulsive: float = 1.0, lw_noise: float = 0.0, lw_coward: float = 0.0,
                 q_min: float = 0.01, pt_thld: float = 0.9, max_eta: float = 4.0, max_n_rep: int = 0,
                 sample_pids: float = 1.0):
        """Condensation loss, dense formulation (oc.py:350-436) without the N x K matrices.

        Args: as ``CondensationLossRG``; ``max_n_rep``: if more repulsive pairs than this are
            inside the unit radius, a random subset of about that size is used (0: all).
        """
        super().__init__()
        self.save_hyperparameters()

    def forward(self, *, beta: T, x: T, particle_id: T, reconstructable: T, pt: T,
                ec_hit_mask: T | None = None, eta: T, **kwargs) -> MultiLossFctReturn:
        return self._forward(beta=beta, x=x, particle_id=particle_id,
                             reconstructable=reconstructable, pt=pt, ec_hit_mask=ec_hit_mask,
                             eta=eta, mask_eta=True)
