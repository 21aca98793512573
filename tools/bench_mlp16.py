"""Micro-benchmark of the bf16 fused MLP kernels on the cfg3 graph (32 events x 2M edges):
relational-model shape (h[tgt], h[src], e -> 40 -> 40 -> 4), fp32 kernel beside it."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import _capi, ops, ops_bf16 as B, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--bwd", action="store_true")
ap.add_argument("--head", action="store_true", help="also time the edge-weight head shape")
args = ap.parse_args()
dev = torch.device("cuda", 0)
events = [synthetic.make_event(100 + i, 150_000, 2_000_000, dev) for i in range(args.events)]
batch = G.collate(events)
del events
gi = ops.graph_index(batch.edge_index, batch.num_nodes)
N, E = batch.num_nodes, gi.n_edges
torch.manual_seed(0)
m = G.MLP(14, 4, 40, L=3).to(dev)
W = [l.weight.detach().contiguous() for l in m.linears()]
b = [l.bias.detach().contiguous() for l in m.linears()]
mlp = ops._fill_mlp(W, b)
h32 = torch.randn(N, 5, device=dev)
e32 = torch.randn(E, 4, device=dev)
h16 = B.empty_rows(N, 5, dev, zero=True)
h16.copy_(h32)
e16 = B.empty_rows(E, 4, dev, zero=True)
e16.copy_(e32)


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / iters


def fwd16():
    return B.mlp_forward_raw([h16, h16, e16], [gi.tgt, gi.src, None], [True, True, True], W, b, n_rows=E,
                             epilogue=_capi.EPI_NONE, ca=0.0, cb=1.0, res=None, out_idx=None, out_rows=E,
                             mlp=mlp)


def fwd32():
    with torch.no_grad():
        return m.fused([ops.Seg(h32, gi.tgt, True, ("tgt", gi)), ops.Seg(h32, gi.src, True, ("src", gi)),
                        ops.Seg(e32, None, True)], n_rows=E)


t16 = timeit(fwd16, args.iters)
t32 = timeit(fwd32, args.iters)
y16 = fwd16().float()
y32 = fwd32()
err = (y16 - y32).abs().max().item()
alg16 = E * (8 + 8 + 8 + 2 * 16)  # ids + e + out + two gathered 16-byte rows
print(f"rel fwd  E={E}: bf16 {t16:.3f} ms ({E / t16 / 1e6:.1f} G rows/s, alg {alg16 / t16 / 1e9:.2f} TB/s) | "
      f"fp32 {t32:.3f} ms | max|bf16-fp32| {err:.3e} (scale {y32.abs().max().item():.2f})")

if args.bwd:
    ge16 = B.empty_rows(E, 4, dev, zero=True)
    ge16.copy_(torch.randn(E, 4, device=dev))
    ga16 = B.empty_rows(N, 4, dev, zero=True)
    ga16.copy_(torch.randn(N, 4, device=dev))

    def bwd16():
        return B.mlp_backward_raw([h16, h16, e16], [gi.tgt, gi.src, None], [True, True, True], W, b, n_rows=E,
                                  epilogue=_capi.EPI_NONE, ca=0.0, cb=1.0, gout=[(ge16, None), (ga16, gi.tgt)],
                                  need_seg=[True, True, True], want_dw=True, mlp=mlp)

    tb = timeit(bwd16, args.iters)
    alg = E * (8 + 8 + 32 + 8 + 8 + 8 + 32)
    print(f"rel bwd  E={E}: bf16 {tb:.3f} ms ({E / tb / 1e6:.1f} G rows/s, alg {alg / tb / 1e9:.2f} TB/s)")


if args.head:
    # ECForGraphTCN.W: h[src], h[tgt], four edge tensors -> 40 -> 40 -> 1, sigmoid epilogue,
    # W scattered back into edge_index order (out_idx = perm)
    mh = G.MLP(26, 1, 40, L=3).to(dev)
    Wh = [l.weight.detach().contiguous() for l in mh.linears()]
    bh = [l.bias.detach().contiguous() for l in mh.linears()]
    mlph = ops._fill_mlp(Wh, bh)
    es = []
    for i in range(4):
        t = B.empty_rows(E, 4, dev, zero=True)
        t.copy_(torch.randn(E, 4, device=dev))
        es.append(t)

    def head(out_idx):
        return B.mlp_forward_raw([h16, h16] + es, [gi.src, gi.tgt, None, None, None, None], [False] * 6, Wh, bh,
                                 n_rows=E, epilogue=_capi.EPI_SIGMOID, ca=0.001, cb=0.998, res=None,
                                 out_idx=out_idx, out_rows=E, mlp=mlph)

    t_sc = timeit(lambda: head(gi.perm), args.iters)
    t_id = timeit(lambda: head(None), args.iters)
    print(f"head fwd E={E}: scattered W {t_sc:.3f} ms | CSR-order W {t_id:.3f} ms")
    if args.bwd:
        gw = torch.randn(E, 1, device=dev)

        def hbwd(gidx_on, spos):
            return B.mlp_backward_raw([h16, h16] + es, [gi.src, gi.tgt, None, None, None, None], [False] * 6, Wh, bh,
                                      n_rows=E, epilogue=_capi.EPI_SIGMOID, ca=0.001, cb=0.998,
                                      gout=[(gw, gi.perm if gidx_on else None)], need_seg=[True] * 6, want_dw=True,
                                      mlp=mlph, gidx=[gi.spos_inv if spos else None] + [None] * 5)

        for a_, b_ in ((True, True), (False, True), (False, False)):
            t = timeit(lambda: hbwd(a_, b_), args.iters)
            print(f"head bwd: gathered g_W {a_}, source-sorted g_h[src] {b_}: {t:.3f} ms")
