#!/usr/bin/env python3
"""Randomised pruned-against-exhaustive comparisons on the GPU (fault / boundary hunting): the sorted-chunk
neighbour search, the DBSCAN radius graph and the spatial condensation-loss passes against their brute-force /
dense forms on random sizes (incl. chunk and batch boundaries), dimensions 1..16, k, radii, event splits.

    python tools/gpu_stress_pruned.py [seed] [rounds]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_tracking_amd import losses_oc, ops, postprocessing, synthetic  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
SIZES = (1, 2, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 5000, 8191, 8192, 8193, 20000)


def cloud(n, d):
    c = g.uniform(-2, 2, size=(max(n // 25, 1), d))
    x = c[g.integers(0, len(c), size=n)] + 0.05 * g.normal(size=(n, d))
    x[::7] = g.uniform(-2, 2, size=(len(x[::7]), d))
    if n > 10:
        x[5::40] = x[4::40][:len(x[5::40])]   # duplicates: ties
    return torch.from_numpy(x.astype(np.float32)).to(dev)


for rnd in range(rounds):
    n = int(g.choice(SIZES)) if rnd % 2 == 0 else int(g.integers(2, 30000))
    d = int(g.integers(1, 17))
    k = int(g.choice((1, 3, 16, 64, 65, 100, 192, 193, 256, 448)))
    r = None if g.random() < 0.3 else float(g.uniform(0.2, 1.5))
    x = cloud(n, d)
    seg = None
    if n > 3 and g.random() < 0.4:
        cuts = np.sort(g.choice(np.arange(1, n), size=min(int(g.integers(1, 6)), n - 1), replace=False))
        seg = torch.tensor([0, *cuts.tolist(), n], dtype=torch.int64, device=dev)
    res = {}
    for flags in (1, 2):
        ops._KNN_FLAGS = flags
        res[flags] = ops.knn_graph(x, k, r, seg_ptr=seg)
    ops._KNN_FLAGS = 0
    assert torch.equal(res[1], res[2]), f"kNN n={n} d={d} k={k} r={r} seg={seg}"
    # radius graph
    eps = float(g.uniform(0.1, 0.6))
    rg = {}
    for flags in (1, 2):
        postprocessing.RADIUS_FLAGS = flags
        fr = postprocessing.DBSCANFastRescan(x, max_eps=eps)
        rg[flags] = (fr._off.clone(), fr._nbr[:fr._n_edges].clone(), fr._dist[:fr._n_edges].clone())
    postprocessing.RADIUS_FLAGS = 0
    assert all(torch.equal(a, b) for a, b in zip(rg[1], rg[2])), f"radius graph n={n} d={d} eps={eps}"
    # condensation losses
    ev = synthetic.make_pileup_event(int(g.integers(1, 1000)), n, dim=d, n_particles=max(n // 12, 2)) if n >= 50 else None
    if ev is not None and bool(((ev["pt"] > 0.9) & (ev["particle_id"] > 0) & (ev["eta"].abs() < 4.0)).any()):
        out = {}
        for mode in ("on", "off"):
            losses_oc.SPATIAL = mode
            b = ev["beta"].to(dev).requires_grad_(True)
            xx = (ev["x"] * 0.5).to(dev).requires_grad_(True)
            cls = losses_oc.CondensationLossRG if rnd % 2 else losses_oc.CondensationLossTiger
            ret = cls(lw_repulsive=2.0, lw_noise=0.5, lw_coward=0.25)(
                beta=b, x=xx, particle_id=ev["particle_id"].to(dev), reconstructable=ev["reconstructable"].to(dev),
                pt=ev["pt"].to(dev), eta=ev["eta"].to(dev))
            ret.loss.backward()
            out[mode] = (float(ret.loss.detach()), xx.grad.clone(), b.grad.clone())
        losses_oc.SPATIAL = "auto"
        la, lb = out["on"][0], out["off"][0]
        if la != la and lb != lb:   # (no noise hit in a tiny event: the noise term is NaN in both, as in the reference)
            print(f"round {rnd}: n={n} d={d}: loss NaN in both forms", flush=True)
            continue
        assert abs(la - lb) <= 1e-5 * abs(lb) + 1e-9, f"OC loss n={n} d={d}: {la} vs {lb}"
        for a, b_ in zip(out["on"][1:], out["off"][1:]):
            assert (a - b_).abs().max().item() <= 1e-4 * max(1.0, b_.abs().max().item()), f"OC grads n={n} d={d}"
    print(f"round {rnd}: n={n} d={d} k={k} r={r} seg={'yes' if seg is not None else 'no'} eps={eps:.2f} ok", flush=True)
print("pruned stress ok seed", seed)
