"""``MLP`` with the reference's constructor, ``state_dict`` keys and semantics
(reference: models/mlp.py:18-62), executed by the fused HIP gather-MLP kernel.

``layers`` is a ``ModuleList`` of ``Linear`` / ``ReLU`` entries in the reference's
order so checkpoints exchange freely (keys ``layers.0.weight``, ``layers.2.weight``,
...).  The Linear/ReLU objects are parameter containers only - ``forward`` never
calls them; it hands their weights to ``ops.fused_mlp``.
"""

from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import Tensor, nn

from . import _capi, ops


class MLP(nn.Module):
    def __init__(self, input_size: int, output_size: int, hidden_dim: int | None, L: int = 3, *,
                 bias: bool = True, include_last_activation: bool = False):
        """Linear -> ReLU -> ... -> Linear with ``L`` linear layers.

        Args:
            input_size: input feature dimension
            output_size: output feature dimension
            hidden_dim: hidden width; ``None`` = ``max(input_size, output_size)``
            L: number of linear layers (the fused kernels cover ``L`` in {2, 3}, the
                only values on the reference's hot path)
            bias: use biases
            include_last_activation: ReLU after the last layer
        """
        super().__init__()
        if hidden_dim is None:
            hidden_dim = max(input_size, output_size)
        widths = [input_size] + [hidden_dim] * (L - 1) + [output_size]
        mods: list[nn.Module] = []
        for i in range(L):
            if i > 0:
                mods.append(nn.ReLU())
            mods.append(nn.Linear(widths[i], widths[i + 1], bias=bias))
        if include_last_activation:
            mods.append(nn.ReLU())
        self.layers = nn.ModuleList(mods)
        self._last_act = include_last_activation
        self.in_dim, self.hidden_dim, self.out_dim, self.L = input_size, hidden_dim, output_size, L

    def reset_parameters(self) -> None:
        for layer in self.layers:
            if hasattr(layer, "reset_parameters"):
                layer.reset_parameters()

    def linears(self) -> list[nn.Linear]:
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    def fused(self, segs: Sequence[ops.Seg], **kw) -> Tensor:
        """Run this MLP on a gathered/concatenated input (see ``ops.fused_mlp``)."""
        lin = self.linears()
        return ops.fused_mlp(segs, [m.weight for m in lin], [m.bias for m in lin], **kw)

    def forward(self, x: Tensor) -> Tensor:
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        epi = _capi.EPI_RELU if self._last_act else _capi.EPI_NONE
        out = self.fused([ops.Seg(x2)], epilogue=epi)
        return out.reshape(*lead, self.out_dim)
