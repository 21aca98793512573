"""DBSCAN post-processing on the device.

Reference: postprocessing/fastrescanner.py:6-66 (``DBSCANFastRescan``: one radius graph at
``max_eps``, then ``cluster(eps, min_pts)`` for many hyperparameters, as the DBSCAN
hyperparameter scanners of postprocessing/dbscanscanner.py:146-187 call it).  Same
constructor, same ``cluster`` signature, same labels (cluster numbering, border points,
noise = -1); the graph and the clustering stay on the GPU, ``cluster`` hands back a numpy
array like the reference (``cluster_device`` the device tensor).  The tracking metrics that
consume the labels are CPU validation code and stay with the caller.
"""

from __future__ import annotations

import os

import numpy as np
import torch
from torch import Tensor

from . import _capi, ops


#: bit 0: pruned radius graph below its size threshold too; bit 1: brute force only (tests, measurements)
RADIUS_FLAGS = int(os.environ.get("GNNTRK_RADIUS_FLAGS", "0"))


class DBSCANFastRescan:
    def __init__(self, x, max_eps: float = 1.0, *, n_jobs: int | None = None, device=None):
        """Args as fastrescanner.py:7-24 (``n_jobs`` is accepted and ignored).  ``x``: the
        cluster coordinates ``[N, D]``, D <= 32 - a device tensor, or a numpy array /
        host tensor that is copied to ``device`` (default ``cuda:0``)."""
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x))
        if not x.is_cuda and device is None and torch.cuda.is_available():
            device = torch.device("cuda", 0)
        if device is not None:
            x = x.to(device)
        _capi.require_device(x)
        if x.dim() != 2:
            raise ValueError("DBSCANFastRescan: x must be [N, D]")
        self.x = ops._as_rows(x.detach().to(torch.float32))
        self._max_eps = float(max_eps)
        self._n_jobs = n_jobs
        self._reset_graph(self._max_eps)

    def _reset_graph(self, max_eps: float) -> None:
        """The radius-neighbourhood graph (fastrescanner.py:25-39): CSR offsets, neighbour
        ids and fp64 distances."""
        lib = _capi.load()
        x = self.x
        n, dim = int(x.shape[0]), int(x.shape[1])
        st = ops._stream(x)
        cnt = torch.empty(max(n, 1), dtype=torch.int32, device=x.device)
        self._off = torch.empty(n + 1, dtype=torch.int64, device=x.device)
        # (the library prunes the N^2 walk with the sorted chunks it builds in ws_p; same output)
        nb = int(lib.gnntrk_radius_points_workspace_bytes(n, dim))
        ws_p = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
        _capi.check(lib.gnntrk_radius_count_ws(ops._p(x), n, dim, ops._row_stride(x), float(max_eps), ops._p(cnt),
                                               ops._p(self._off), ops._p(ws_p), nb, RADIUS_FLAGS, st), lib)
        m = int(self._off[n].item())
        self._nbr = torch.empty(max(m, 1), dtype=torch.int32, device=x.device)
        self._dist = torch.empty(max(m, 1), dtype=torch.float64, device=x.device)
        ne = int(lib.gnntrk_radius_edges_workspace_bytes(m)) if nb else 0
        ws_e = torch.empty(ne, dtype=torch.uint8, device=x.device) if ne else None
        _capi.check(lib.gnntrk_radius_fill_ws(ops._p(x), n, dim, ops._row_stride(x), float(max_eps),
                                              ops._p(self._off), m, ops._p(self._nbr), ops._p(self._dist),
                                              ops._p(ws_p), nb, ops._p(ws_e), ne, RADIUS_FLAGS, st), lib)
        self._n_edges = m
        self._max_eps = float(max_eps)

    def cluster_device(self, eps: float = 1.0, min_pts: int = 1) -> Tensor:
        """Labels as an int64 device tensor."""
        if eps > self._max_eps:
            self._reset_graph(eps)
        lib = _capi.load()
        x = self.x
        n = int(x.shape[0])
        dev = x.device
        st = ops._stream(x)
        core = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
        root = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        changed = torch.ones(1, dtype=torch.int32, device=dev)
        labels = torch.empty(n, dtype=torch.int64, device=dev)
        args = (ops._p(self._off), ops._p(self._nbr), ops._p(self._dist), n, float(eps))
        _capi.check(lib.gnntrk_dbscan_init(args[0], args[2], n, float(eps), int(min_pts), ops._p(core),
                                           ops._p(root), st), lib)
        # a few rounds per host check (compact clusters need one or two), more per check for
        # long chains; n rounds always suffice (the lowest index moves at least one hop per round)
        rounds, done = 4, 0
        while True:
            _capi.check(lib.gnntrk_dbscan_propagate(*args, ops._p(core), ops._p(root), rounds, ops._p(changed), st),
                        lib)
            done += rounds
            if int(changed.item()) == 0:
                break
            if done > n + 8:
                raise RuntimeError("DBSCAN label propagation did not converge")
            rounds = min(2 * rounds, 64)
        ws = torch.empty(max(int(lib.gnntrk_dbscan_workspace_bytes(n)), 256), dtype=torch.uint8, device=dev)
        n_clusters = torch.empty(1, dtype=torch.int64, device=dev)
        _capi.check(lib.gnntrk_dbscan_labels(*args, ops._p(core), ops._p(root), ops._p(labels), ops._p(n_clusters),
                                             ops._p(ws), ws.numel(), st), lib)
        return labels

    def cluster(self, eps: float = 1.0, min_pts: int = 1) -> np.ndarray:
        """fastrescanner.py:41-66: DBSCAN labels (``np.intp``), noise = -1."""
        return self.cluster_device(eps, min_pts).cpu().numpy().astype(np.intp)


def dbscan(x, eps: float, min_samples: int, device=None) -> np.ndarray:
    """``sklearn.cluster.DBSCAN(eps, min_samples).fit_predict(x)`` on the device."""
    return DBSCANFastRescan(x, max_eps=eps, device=device).cluster(eps, min_samples)
