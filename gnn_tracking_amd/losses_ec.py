"""Edge-classification losses (reference: metrics/losses/ec.py:13-183)."""

from __future__ import annotations

import math

import torch
from torch import Tensor

from . import ops
from .hparams import HyperparametersMixin


def falsify_low_pt_edges(*, y: Tensor, edge_index: Tensor | None = None, pt: Tensor | None = None,
                         pt_thld: float = 0.0) -> Tensor:
    """Edges whose first hit has ``pt <= pt_thld`` count as false (ec.py:71-92).
    Index bookkeeping on labels only; ``EdgeWeightBCELoss`` folds it into its kernel."""
    if math.isclose(pt_thld, 0.0):
        return y
    assert edge_index is not None
    assert pt is not None
    return y.bool() & (pt[edge_index[0, :]] > pt_thld)


class EdgeWeightBCELoss(torch.nn.Module, HyperparametersMixin):
    """Binary cross entropy of the edge weights (ec.py:95-121), one fused reduction."""

    def __init__(self, *, pt_thld: float = 0.0):
        super().__init__()
        self.save_hyperparameters()

    def forward(self, *, w: Tensor, y: Tensor, edge_index: Tensor | None = None,
                pt: Tensor | None = None, **kwargs) -> Tensor:
        return ops.bce_loss(w, y, edge_index, pt, float(self.hparams.pt_thld))


def binary_focal_loss(*, inpt: Tensor, target: Tensor, alpha: float = 0.25, gamma: float = 2.0,
                      pos_weight: Tensor | float | None = None) -> Tensor:
    """Binary focal loss, https://arxiv.org/abs/1708.02002 (ec.py:35-68), one fused reduction.
    ``pos_weight``: a scalar (or one-element tensor)."""
    pw = 1.0 if pos_weight is None else float(torch.as_tensor(pos_weight).reshape(-1)[0])
    return ops.focal_loss(inpt, target, alpha=alpha, gamma=gamma, pos_weight=pw)


class EdgeWeightFocalLoss(torch.nn.Module, HyperparametersMixin):
    """Focal loss of the edge weights against the pt-falsified labels (ec.py:124-150)."""

    def __init__(self, *, alpha=0.25, gamma=2.0, pos_weight=None, pt_thld: float = 0.0):
        super().__init__()
        self.save_hyperparameters()

    def forward(self, *, w: Tensor, y: Tensor, edge_index: Tensor | None = None,
                pt: Tensor | None = None, **kwargs) -> Tensor:
        pw = self.hparams.pos_weight
        pw = 1.0 if pw is None else float(torch.as_tensor(pw).reshape(-1)[0])
        return ops.focal_loss(w, y, edge_index, pt, float(self.hparams.pt_thld), alpha=self.hparams.alpha,
                              gamma=self.hparams.gamma, pos_weight=pw)


class HaughtyFocalLoss(torch.nn.Module, HyperparametersMixin):
    """Focal loss whose positive term is weighted by the pt-falsified label (ec.py:153-183)."""

    def __init__(self, *, alpha: float = 0.25, gamma: float = 2.0, pt_thld=0.0):
        super().__init__()
        self.save_hyperparameters()

    def forward(self, *, w: Tensor, y: Tensor, edge_index: Tensor, pt: Tensor, **kwargs) -> Tensor:
        return ops.focal_loss(w, y, edge_index, pt, float(self.hparams.pt_thld), alpha=self.hparams.alpha,
                              gamma=self.hparams.gamma, haughty=True)
