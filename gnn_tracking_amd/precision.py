"""Storage-precision switch of the hot path.

fp32 (default) is the mode pinned against the reference's golden vectors.  The bf16 mode
(BASELINE configs 3/4: "bf16 storage for x, e, e~, aggr and the MFMA inputs, fp32
accumulate") is entered either explicitly::

    with gnn_tracking_amd.bf16_storage():
        out = model(data)

or implicitly under ``torch.autocast("cuda", dtype=torch.bfloat16)`` - which is what
Lightning's ``Trainer(precision="bf16-mixed")`` wraps the reference's training step in
(training/base.py).  Parameters and their gradients stay fp32 in both modes.
"""

from __future__ import annotations

import contextlib

import torch

_FORCED = False


@contextlib.contextmanager
def bf16_storage(enabled: bool = True):
    global _FORCED
    old, _FORCED = _FORCED, bool(enabled)
    try:
        yield
    finally:
        _FORCED = old


def use_bf16() -> bool:
    if _FORCED:
        return True
    try:
        return bool(torch.is_autocast_enabled()) and torch.get_autocast_gpu_dtype() == torch.bfloat16
    except Exception:  # pragma: no cover - torch builds without the autocast query
        return False
