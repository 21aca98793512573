#!/usr/bin/env python3
"""Ablation timings of the fused gather-MLP kernels on the relational-model shape
(14 -> 40 -> 40 -> 4, rows = edges in CSR order) via the C ABI's debug_flags.

    python tools/ablate_mlp.py [n_events]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import _capi, ops, synthetic  # noqa: E402

n_ev = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
lib = _capi.load()
evs = [synthetic.make_event(100 + i, 150_000, 2_000_000, dev) for i in range(n_ev)]
b = G.collate(evs)
del evs
gi = ops.graph_index(b.edge_index, b.num_nodes)
N, E = b.num_nodes, b.num_edges
torch.manual_seed(0)
h = torch.randn(N, 5, device=dev)
e = torch.randn(E, 4, device=dev)
W = [torch.randn(40, 14, device=dev) * 0.3, torch.randn(40, 40, device=dev) * 0.2,
     torch.randn(4, 40, device=dev) * 0.2]
bb = [torch.randn(40, device=dev) * 0.1, torch.randn(40, device=dev) * 0.1,
      torch.randn(4, device=dev) * 0.1]
p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
stream = torch.cuda.current_stream().cuda_stream


def mk_mlp():
    return _capi.make_mlp([p(w) for w in W], [p(x) for x in bb], 14, 40, 4)


def segs(a):
    a.n_seg = 3
    a.seg[0] = _capi.Seg(p(h), p(gi.tgt), 5, 5, 1, 0)
    a.seg[1] = _capi.Seg(p(h), p(gi.src), 5, 5, 1, 0)
    a.seg[2] = _capi.Seg(p(e), None, 4, 4, 1, 0)


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = torch.empty(E, 4, device=dev)
print(f"E={E} N={N}")
for flags in (0, 1, 2, 3):
    a = _capi.MlpFwdArgs()
    a.mlp = mk_mlp()
    segs(a)
    a.epilogue, a.n_rows, a.ca, a.cb = 0, E, 0.0, 1.0
    a.out, a.out_stride, a.debug_flags = p(out), 4, flags
    t = timeit(lambda: _capi.check(lib.gnntrk_mlp_forward(C.byref(a), stream)))
    print(f"fwd flags={flags}: {t:8.3f} ms  ({4640*E/t/1e9:6.1f} TF alg)")

g_e = torch.randn(E, 4, device=dev)
g_aggr = torch.randn(N, 4, device=dev)
gxi = torch.empty(E, 5, device=dev)
gxj = torch.empty(E, 5, device=dev)
ge = torch.empty(E, 4, device=dev)
gW = [torch.empty_like(w) for w in W]
gb = [torch.empty_like(x) for x in bb]
mlp = mk_mlp()
ws = torch.empty(lib.gnntrk_mlp_backward_workspace_bytes(C.byref(mlp)), dtype=torch.uint8, device=dev)
for flags, dw, ng in ((0, 1, 2), (0, 1, 1), (1, 1, 2), (2, 1, 2), (3, 1, 2), (8, 1, 2), (11, 1, 2), (0, 0, 2),
                      (3, 0, 2)):
    a = _capi.MlpBwdArgs()
    a.mlp = mk_mlp()
    segs(a)
    a.epilogue, a.n_rows, a.ca, a.cb = 0, E, 0.0, 1.0
    a.n_gout = ng
    a.gout[0] = _capi.GTerm(p(g_e), None, 4, 0)
    a.gout[1] = _capi.GTerm(p(g_aggr), p(gi.tgt), 4, 0)
    a.gseg[0] = _capi.GSeg(p(gxi), None, 5, 0)
    a.gseg[1] = _capi.GSeg(p(gxj), None, 5, 0)
    a.gseg[2] = _capi.GSeg(p(ge), None, 4, 0)
    if dw:
        for i in range(3):
            a.gW[i] = p(gW[i])
            a.gb[i] = p(gb[i])
    a.debug_flags = flags
    t = timeit(lambda: _capi.check(lib.gnntrk_mlp_backward(C.byref(a), p(ws), ws.numel(), stream)))
    print(f"bwd flags={flags:2d} dW={dw} n_gout={ng}: {t:8.3f} ms  ({3*4640*E/t/1e9:6.1f} TF alg)")
