"""Edge-classification losses (reference: metrics/losses/ec.py:71-121)."""

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
