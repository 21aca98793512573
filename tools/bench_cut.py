"""Threshold cut / orphan masking at the cfg3 size (E = 64 M edges, N = 4.8 M nodes):
the stream-compaction kernels (csrc/compact.hip) beside the torch device ops the
reference's formulation maps to (boolean indexing, sort-based unique, index relabel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tracking_amd import graph_cut  # noqa: E402

dev = torch.device("cuda", 0)
E, N = 64_000_000, 4_800_000
torch.manual_seed(0)
w = torch.rand(E, device=dev)
ei = torch.randint(0, N * 3 // 4, (2, E // 2), device=dev)  # the graph after a 50 % cut


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / iters


def torch_cut():
    m = w > 0.5
    return m, torch.nonzero(m).reshape(-1)


def torch_orphans():
    conn = ei.flatten().unique()
    hit = torch.zeros(N, dtype=torch.bool, device=dev)
    hit[conn] = True
    relabel = torch.full((N,), -1, dtype=torch.long, device=dev)
    relabel[conn] = torch.arange(conn.numel(), device=dev)
    return hit, conn, relabel[ei]


m0, i0 = torch_cut()
m1, i1 = graph_cut.threshold_compact(w, 0.5)
assert torch.equal(m0, m1) and torch.equal(i0, i1.long())
h0, c0, e0 = torch_orphans()
h1, c1, e1 = graph_cut.connected_nodes(ei, N)
assert torch.equal(h0, h1) and torch.equal(c0, c1.long()) and torch.equal(e0, e1)
ta, tb = timeit(lambda: graph_cut.threshold_compact(w, 0.5)), timeit(torch_cut)
# algorithmic bytes: read w twice (count + write passes), mask 1 B, 4 B per kept index
alg = E * (4 + 4 + 1) + i1.numel() * 4
print(f"threshold cut  E={E}: kernel {ta:.3f} ms ({alg / ta / 1e9:.2f} TB/s algorithmic) | torch mask+nonzero {tb:.3f} ms")
ta, tb = timeit(lambda: graph_cut.connected_nodes(ei, N)), timeit(torch_orphans)
m = ei.numel()
alg = m * 8 + m * 1 + N * (1 + 1 + 4 + 1) + c1.numel() * 4 + m * (8 + 4 + 8)
print(f"orphan masking 2E'={m} N={N}: kernel {ta:.3f} ms ({alg / ta / 1e9:.2f} TB/s algorithmic) | "
      f"torch unique+relabel {tb:.3f} ms")
