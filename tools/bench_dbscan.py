#!/usr/bin/env python3
"""DBSCAN fast rescan at the cfg5 size: radius graph of a 200k-hit, 8-D pile-up-like cloud at
max_eps, then several (eps, min_pts) rescans - device kernels beside sklearn on the host
(the calls postprocessing/fastrescanner.py makes: NearestNeighbors.radius_neighbors +
dbscan_inner), labels compared bit for bit."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_knn import cloud  # noqa: E402
from gnn_tracking_amd.postprocessing import DBSCANFastRescan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
max_eps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
trials = ((max_eps, 1), (0.6 * max_eps, 2), (0.4 * max_eps, 3), (0.3 * max_eps, 4))
x = cloud(500, n)
xd = x.cuda()
DBSCANFastRescan(xd[:4096], max_eps=max_eps).cluster(max_eps, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
fr = DBSCANFastRescan(xd, max_eps=max_eps)
torch.cuda.synchronize()
t_graph = time.perf_counter() - t0
labels = {}
t0 = time.perf_counter()
for eps, mp in trials:
    labels[(eps, mp)] = fr.cluster_device(eps, mp)
torch.cuda.synchronize()
t_clu = (time.perf_counter() - t0) / len(trials)
m = fr._n_edges
print(f"GPU  n={n} D=8 max_eps={max_eps}: radius graph {t_graph*1e3:.1f} ms ({m} edges, "
      f"{2 * n * n * 8 * 3 / t_graph / 1e12:.2f} fp64 Tflop/s over both passes), "
      f"rescan {t_clu*1e3:.2f} ms per (eps, min_pts)")

from sklearn.cluster._dbscan_inner import dbscan_inner  # noqa: E402
from sklearn.neighbors import NearestNeighbors  # noqa: E402

xn = x.numpy()
t0 = time.perf_counter()
nm = NearestNeighbors(radius=max_eps, n_jobs=-1).fit(xn)
dist, ind = nm.radius_neighbors(xn, radius=max_eps, return_distance=True)
t_sk_graph = time.perf_counter() - t0
t0 = time.perf_counter()
ok = True
for eps, mp in trials:
    neigh = np.empty(n, dtype=object)
    for i in range(n):
        neigh[i] = ind[i][dist[i] <= eps]
    core = np.asarray([len(v) >= mp for v in neigh], dtype=np.uint8)
    lab = np.full(n, -1, dtype=np.intp)
    dbscan_inner(core, neigh, lab)
    ok = ok and np.array_equal(lab, labels[(eps, mp)].cpu().numpy())
t_sk_clu = (time.perf_counter() - t0) / len(trials)
print(f"host sklearn ({os.cpu_count()} cpus, n_jobs=-1): radius graph {t_sk_graph:.2f} s, rescan {t_sk_clu:.2f} s per "
      f"trial; labels identical: {ok}")
