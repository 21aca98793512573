"""Seeded synthetic TrackML-shaped hit graphs (SURVEY.md section 8d generator spec).

There is no network for datasets, so bench/tests use graphs whose SHAPE statistics
follow the graphs the reference's GraphBuilder produces from TrackML events: 14 node
features with the measured value ranges, 4 edge features (dr, dphi, dz, dR), mean degree
E/N, edges between hits that are close in phi, both edge directions, shuffled COO
order, >=3 % isolated nodes, true-edge fraction 0.31, int64 particle ids up to 2^60 with
4 % noise (id 0).  Node ids are NOT phi-sorted (worst case for gather locality).
"""

from __future__ import annotations

import math

import torch

from .data import Data


def make_event(seed: int, n_nodes: int, n_edges: int, device="cpu", *, node_dim: int = 14,
               isolated_frac: float = 0.03, max_offset: int = 64, phi_sorted_ids: bool = False) -> Data:
    """``phi_sorted_ids``: renumber the hits by phi, so that the endpoints of an edge have
    neighbouring ids (what a dataset written out in detector order looks like: the best case for
    gather locality).  Default False: ids are random with respect to the geometry (worst case)."""
    assert node_dim >= 6 and n_edges % 2 == 0
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    U = lambda *s: torch.rand(*s, generator=g, device=dev)  # noqa: E731
    N, E = n_nodes, n_edges
    x = torch.empty(N, node_dim, device=dev)
    x[:, 0] = 0.03 + 0.15 * U(N)                       # r
    x[:, 1] = 2 * U(N) - 1                             # phi / pi
    x[:, 2] = 3 * U(N) - 1.5                           # z
    x[:, 3] = (2 * torch.randn(N, generator=g, device=dev)).clamp(-4.6, 4.6)  # eta
    x[:, 4:6] = 66 * U(N, 2) - 33                      # u, v
    x[:, 6:] = 1.5 * U(N, node_dim - 6)

    # live (non-isolated) nodes, ordered by phi
    n_iso = int(math.ceil(isolated_frac * N))
    shuffled = torch.randperm(N, generator=g, device=dev)
    live = shuffled[n_iso:]
    live = live[torch.argsort(x[live, 1])]
    L = live.numel()
    half = E // 2
    i = torch.randint(0, L, (half,), generator=g, device=dev)
    off = torch.randint(1, max_offset + 1, (half,), generator=g, device=dev)
    sign = torch.randint(0, 2, (half,), generator=g, device=dev) * 2 - 1
    j = (i + sign * off) % L
    a, b = live[i], live[j]
    src = torch.cat([a, b])
    tgt = torch.cat([b, a])
    shuf = torch.randperm(E, generator=g, device=dev)
    edge_index = torch.stack([src[shuf], tgt[shuf]]).contiguous()

    s, t = edge_index[0], edge_index[1]
    dr = x[s, 0] - x[t, 0]
    dphi = x[s, 1] - x[t, 1]
    dphi = dphi - 2 * torch.round(dphi / 2)            # wrap (phi is in units of pi)
    dz = x[s, 2] - x[t, 2]
    dR = torch.sqrt((x[s, 3] - x[t, 3]) ** 2 + dphi ** 2)
    edge_attr = torch.stack([dr, dphi, dz, dR], dim=1).contiguous()

    y = U(E) < 0.31
    pid = torch.randint(2 ** 52, 2 ** 60, (N,), generator=g, device=dev, dtype=torch.int64)
    pid[U(N) < 0.04] = 0
    pt = torch.exp(0.7 * torch.randn(N, generator=g, device=dev))
    if phi_sorted_ids:
        order = torch.argsort(x[:, 1])                 # new id -> old id
        rank = torch.empty_like(order)
        rank[order] = torch.arange(N, device=dev)      # old id -> new id
        x, pt, pid = x[order].contiguous(), pt[order].contiguous(), pid[order].contiguous()
        edge_index = rank[edge_index].contiguous()
    return Data(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y, pt=pt, particle_id=pid,
                eta=x[:, 3].clone(), reconstructable=torch.ones(N, device=dev))


def make_pileup_cloud(seed: int, n_hits: int, dim: int = 8, *, n_clusters: int = 6000,
                      sigma: float = 0.05, noise_frac: float = 0.1) -> torch.Tensor:
    """Latent-space hits of a pile-up-like event: Gaussian
    clusters (sigma) around centres uniform in a radius-3 ball plus uniform noise hits, in
    shuffled order.  numpy generator: the same cloud on every host."""
    import numpy as np

    g = np.random.default_rng(seed)

    def ball(m):
        v = g.normal(size=(m, dim))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v * (3.0 * g.random((m, 1)) ** (1.0 / dim))

    centers = ball(n_clusters)
    n_noise = int(noise_frac * n_hits)
    which = g.integers(0, n_clusters, size=n_hits - n_noise)
    pts = centers[which] + sigma * g.normal(size=(n_hits - n_noise, dim))
    x = np.concatenate([pts, ball(n_noise)]).astype(np.float32)
    return torch.from_numpy(x[g.permutation(n_hits)])


def make_pileup_event(seed: int, n_hits: int, dim: int = 8, *, n_particles: int = 14000,
                      n_clusters: int = 6000) -> dict:
    """Config-5 inputs of the condensation losses on the cloud above: int64 particle ids
    (10 % noise hits with id 0), per-particle pt (log-normal: roughly a seventh of the
    particles pass the 0.9 GeV cut, K of a few thousand at 200 k hits), eta, beta ~ U(0.01, 0.99).
    Returns CPU tensors keyed like the loss's keyword arguments."""
    import numpy as np

    g = np.random.default_rng(seed)
    x = make_pileup_cloud(seed, n_hits, dim, n_clusters=n_clusters)
    pid = torch.from_numpy(g.integers(1, n_particles + 1, size=n_hits)).long() * (2 ** 40)
    pid[torch.from_numpy(g.random(n_hits) < 0.1)] = 0
    pt_of = torch.from_numpy(np.exp(g.normal(-0.5, 0.9, size=n_particles + 1))).float()
    pt = pt_of[pid // 2 ** 40]
    eta = torch.from_numpy(g.normal(0, 2, size=n_hits)).float().clamp(-4.6, 4.6)
    beta = torch.from_numpy(g.uniform(0.01, 0.99, size=n_hits)).float()
    return dict(beta=beta, x=x, particle_id=pid, pt=pt, eta=eta, reconstructable=torch.ones(n_hits))
