"""One optimisation step of the hot path without Lightning.

Reference: ``training/base.py:72-116`` (``TrackingModule``: ``data_preproc`` -> ``forward`` ->
``get_losses`` -> backward -> ``configure_optimizers``' Adam), ``training/ec.py:25-53``
(``ECModule``), ``training/tc.py:20-84`` (``TCModule``) and ``training/ml.py:25-78`` (``MLModule``).  The Lightning trainer around them
(logging, checkpointing, schedulers, OOM tolerance) is control plane and stays in the
reference; what is here is the arithmetic of one step, under the reference's method names, so
that bench.py, the parity tests and a user's own loop run the same code:

    step = TCModule(model=GraphTCN(...), loss_fct=CondensationLossRG(...),
                    preproc=MLGraphConstruction(...))
    loss = step.optimisation_step(batch)

Data parallelism: pass ``flat=dist.FlatParameters(model)``; gradients are then accumulated in
the flat bucket and all-reduced over RCCL before the optimizer step (dist.py).
"""

from __future__ import annotations

from typing import Any, Callable, Optional

import torch
from torch import Tensor, nn

from . import ops
from .precision import bf16_storage


class TrackingModule:
    """``training/base.py:72-116`` without the Lightning base class."""

    def __init__(self, model: nn.Module, *, optimizer: Callable[..., torch.optim.Optimizer] = torch.optim.Adam,
                 scheduler: Optional[Callable[..., Any]] = torch.optim.lr_scheduler.ConstantLR,
                 preproc: Optional[nn.Module] = None, flat=None, bf16: bool = False):
        """
        Args:
            model: the network (``forward(data) -> dict``)
            optimizer: called with the parameters, as Lightning calls ``OptimizerCallable``
                (default Adam, base.py:77); built on first use
            scheduler: called with the optimizer (default ``ConstantLR``, base.py:78 - note that
                its default factor 1/3 applies from construction on, i.e. the reference's first
                five epochs run at a third of the learning rate); ``None`` for none.  Stepping
                it (once per epoch in Lightning) is the caller's business: ``self.scheduler``
            preproc: optional module applied to every batch first (``MLGraphConstruction``)
            flat: ``dist.FlatParameters`` of ``model`` (+ ``preproc`` parameters if trained): one
                gradient bucket, all-reduced before the optimizer step
            bf16: run forward / backward in bf16 storage mode (Lightning ``precision="bf16-mixed"``)
        """
        self.model, self.preproc, self.flat, self.bf16 = model, preproc, flat, bool(bf16)
        self._optimizer_fn, self._optimizer = optimizer, None
        self._scheduler_fn, self.scheduler = scheduler, None

    # -- base.py:94-104
    def forward(self, data, _preprocessed: bool = False):
        if not _preprocessed:
            data = self.data_preproc(data)
        return self.model(data)

    __call__ = forward

    def data_preproc(self, data):
        if self.preproc is not None:
            return self.preproc(data)
        return data

    def parameters(self):
        seen = set()
        for m in (self.model, self.preproc):
            if m is None:
                continue
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    yield p

    # -- base.py:106-112
    def configure_optimizers(self) -> torch.optim.Optimizer:
        if self._optimizer is None:
            params = [self.flat.flat_param] if self.flat is not None else list(self.parameters())
            self._optimizer = self._optimizer_fn(params)
            if self._scheduler_fn is not None:
                self.scheduler = self._scheduler_fn(self._optimizer)
        return self._optimizer

    def get_losses(self, out: dict[str, Any], data):
        raise NotImplementedError

    def _loss(self, data) -> Tensor:
        r = self.training_step(data, 0)
        return r[0] if isinstance(r, tuple) else r

    def zero_grad(self) -> None:
        if self.flat is not None:
            self.flat.zero_grad()
        else:
            self.configure_optimizers().zero_grad(set_to_none=False)

    def backward_step(self, data, *, scale: float = 1.0) -> Tensor:
        """forward + loss + backward of one (micro-)batch; gradients ACCUMULATE."""
        with bf16_storage(self.bf16):
            loss = self._loss(data)
            # (a plain backward of this module's own loss: parameters re-homed by dist.FlatParameters may
            #  take their gradients in place from the backward launches, see ops.grad_sinks_armed)
            with ops.grad_sinks_armed():
                (loss if scale == 1.0 else loss * scale).backward()
        return loss.detach()

    def optimisation_step(self, data) -> Tensor:
        """What one ``Trainer`` iteration does to the parameters (automatic optimisation):
        zero the gradients, ``training_step``, backward, (all-reduce,) optimizer step."""
        opt = self.configure_optimizers()
        self.zero_grad()
        loss = self.backward_step(data)
        if self.flat is not None:
            self.flat.all_reduce_grads()
        opt.step()
        return loss


class ECModule(TrackingModule):
    """Edge-classifier training (``training/ec.py:25-53``)."""

    def __init__(self, model: nn.Module, *, loss_fct: nn.Module, **kwargs):
        super().__init__(model, **kwargs)
        self.loss_fct = loss_fct

    def get_losses(self, out: dict[str, Any], data) -> Tensor:
        # (the reference passes ``data.y.float()``; this package's losses also take the dataset's
        # 1-byte bool labels as they are and skip the conversion pass)
        y = data.y if data.y.dtype in (torch.bool, torch.uint8) else data.y.float()
        return self.loss_fct(w=out["W"], y=y, pt=data.pt, edge_index=data.edge_index)

    def training_step(self, batch, batch_idx: int = 0) -> Tensor:
        batch = self.data_preproc(batch)
        out = self(batch, _preprocessed=True)
        return self.get_losses(out, batch)


class TCModule(TrackingModule):
    """Object-condensation training (``training/tc.py:20-84``): ``loss_fct`` is a
    ``MultiLossFct`` (``CondensationLossRG`` / ``CondensationLossTiger``)."""

    def __init__(self, model: nn.Module, *, loss_fct: nn.Module, **kwargs):
        super().__init__(model, **kwargs)
        self.loss_fct = loss_fct

    def get_losses(self, out: dict[str, Any], data, *, metrics: bool = True):
        losses = self.loss_fct(x=out["H"], particle_id=data.particle_id, beta=out["B"], pt=data.pt,
                               reconstructable=data.reconstructable, eta=data.eta,
                               ec_hit_mask=out.get("ec_hit_mask"), batch=getattr(data, "batch", None),
                               true_edge_index=getattr(data, "true_edges", None))
        if not metrics:  # (the float conversions below synchronise with the device)
            return losses.loss, {}
        m = dict(losses.loss_dct)
        m.update({k + "_weighted": float(v) for k, v in losses.weighted_losses.items()})
        m.update({k: float(v) for k, v in losses.extra_metrics.items()})
        m["total"] = float(losses.loss)
        return losses.loss, m

    def training_step(self, data, batch_idx: int = 0, *, metrics: bool = False):
        data = self.data_preproc(data)
        out = self(data, _preprocessed=True)
        return self.get_losses(out, data, metrics=metrics)


class MLModule(TrackingModule):
    """Metric-learning training (``training/ml.py:25-78``): ``model(data) -> {"H": ...}`` (e.g.
    ``GraphConstructionFCNN``), ``loss_fct`` a ``GraphConstructionHingeEmbeddingLoss``.  The
    scanner of the validation step (``gc_scanner``: k-scans + figures of merit) is validation-only
    control plane; its device part is ``graph_construction.knn_scan``."""

    def __init__(self, model: nn.Module, *, loss_fct: nn.Module, **kwargs):
        super().__init__(model, **kwargs)
        self.loss_fct = loss_fct

    def get_losses(self, out: dict[str, Any], data, *, metrics: bool = True):
        if not hasattr(data, "true_edge_index"):
            # (ml.py:44-47: the point-cloud data saved the true edges as edge_index)
            data.true_edge_index = data.edge_index
        losses = self.loss_fct(x=out["H"], particle_id=data.particle_id, batch=getattr(data, "batch", None),
                               true_edge_index=data.true_edge_index, pt=data.pt, eta=data.eta,
                               reconstructable=data.reconstructable)
        if not metrics:
            return losses.loss, {}
        m = dict(losses.loss_dct)
        m.update({k + "_weighted": float(v) for k, v in losses.weighted_losses.items()})
        m.update({k: float(v) for k, v in losses.extra_metrics.items()})
        m["total"] = float(losses.loss)
        return losses.loss, m

    def training_step(self, batch, batch_idx: int = 0, *, metrics: bool = False):
        batch = self.data_preproc(batch)
        out = self(batch, _preprocessed=True)
        return self.get_losses(out, batch, metrics=metrics)
