"""Event-level data parallelism: one process per GPU, independent events per rank, one
RCCL all-reduce of the flat fp32 gradient buffer per step.

The reference has no distributed code (its multi-GPU story is whatever Lightning's DDP
does); the parameter vector of the edge classifier is 7-71 KB, so the collective is
latency-bound over xGMI: a single in-place all-reduce on one contiguous bucket, no
overlap machinery.
"""

from __future__ import annotations

import os
from typing import Iterable, Sequence

import torch
import torch.distributed as dist
from torch import nn


def init_process_group_from_env(backend: str | None = None, timeout_s: float = 180.0) -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from torchrun's env; initialises the group when
    world_size > 1 (backend "nccl" IS RCCL on ROCm; "gloo" for the CPU tests).

    The rendezvous and the first collective are bounded by ``timeout_s``: a rank that cannot join
    (RCCL initialisation failure, a peer that died, an unreachable master) raises with the
    library's message instead of waiting in a barrier for ever."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        import datetime

        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=timeout_s), **kw)
            # one tiny all-reduce NOW: communicator set-up errors (RCCL: IPC handles, xGMI topology)
            # surface here, with the backend's own message, not in the middle of the first step
            probe = torch.ones(1, device=torch.device("cuda", local) if backend == "nccl" else "cpu")
            dist.all_reduce(probe)
            if int(probe.item()) != world:
                raise RuntimeError(f"all-reduce over {world} ranks returned {probe.item()}")
        except Exception as e:
            raise RuntimeError(f"rank {rank}/{world}: process group ({backend}) could not be set up within "
                               f"{timeout_s:.0f} s: {type(e).__name__}: {e}") from e
    return rank, local, world


class FlatParameters:
    """Re-homes all parameters (and their gradients) of a module into two contiguous
    fp32 buffers so that one all-reduce and one optimizer kernel cover the model.

    ``grad_sink`` (default on): the parameters are marked so that the fused MLP backward launches MAY add
    their weight / bias gradients into these persistent buffers themselves (``ops._param_grad_sinks``)
    instead of returning six tensors per MLP for autograd to ``add_`` - bit-identical sums, 100 fewer
    tiny kernels per step of the edge classifier.  The shortcut bypasses ``AccumulateGrad`` (tensor hooks,
    post-accumulate-grad hooks and DDP's reducer hooks would not fire), so it is only taken inside
    ``ops.grad_sinks_armed()`` - which ``training.TrackingModule.backward_step`` puts around its own plain
    ``loss.backward()`` - and only for parameters without hooks.  Any other ``backward()`` /
    ``torch.autograd.grad`` on these parameters takes the ordinary autograd path."""

    def __init__(self, module: nn.Module, grad_sink: bool = True):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev = params[0].device
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            if grad_sink:
                p._gnntrk_grad_sink = True
            off += k
        self.params = params
        #: a single leaf covering every parameter, for ``torch.optim.*([flat_param])``
        self.flat_param = nn.Parameter(self.flat)
        self.flat_param.grad = self.grad

    def zero_grad(self) -> None:
        self.grad.zero_()

    def all_reduce_grads(self, average: bool = True) -> None:
        """Sum (average) the gradient bucket over all ranks; no-op for one process."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
        if average:
            self.grad.div_(dist.get_world_size())


def shard_events(sizes: Sequence[int], world_size: int) -> list[list[int]]:
    """Greedy size-balanced assignment of events to ranks (largest first): returns for
    every rank the list of event indices it owns.  Events are independent graphs, so
    this is the only partitioning step; there is no data-path collective."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    load = [0] * world_size
    out: list[list[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += sizes[i]
    for r in range(world_size):
        out[r].sort()
    return out
