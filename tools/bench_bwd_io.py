"""Relational / object / encoder backward kernels of the bf16 path at the cfg3 size: buffer-addressed
I/O (wave-uniform descriptors) against the generic per-lane I/O (debug_flags & 128), alternating in
one process, outputs compared bit for bit.

    python tools/bench_bwd_io.py [--events 32] [--iters 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import _capi, ops, ops_bf16 as B, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--only", type=int, default=None, help="0: buffer form only, 128: generic only (for rocprofv3)")
ap.add_argument("--seq-ids", action="store_true",
                help="edge k joins nodes (k / 13, k / 13 + 1) and the source order is the CSR order: no random "
                     "gather / scatter - what the kernels cost without the memory system's share")
ap.add_argument("--node-ids", default="random", choices=("random", "phi"),
                help="phi: hits numbered by phi (the generator's best case for the gathers / the permuted store)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
batch = G.collate([synthetic.make_event(100 + i, 150_000, 2_000_000, dev, phi_sorted_ids=args.node_ids == "phi")
                   for i in range(args.events)])
gi = ops.graph_index(batch.edge_index, batch.num_nodes)
N, E = batch.num_nodes, gi.n_edges
if args.seq_ids:
    k = torch.arange(E, device=dev, dtype=torch.int32)
    gi.tgt = (k // 14).clamp_(max=N - 1).contiguous()
    gi.src = ((k // 14) + 1).clamp_(max=N - 1).contiguous()
    gi.spos_inv = k.clone()
torch.manual_seed(0)


def rows(n, d):
    t = B.empty_rows(n, d, dev, zero=True)
    t.copy_(torch.randn(n, d, device=dev))
    return t


def params(i, o):
    m = G.MLP(i, o, 40, L=3).to(dev)
    return ([l.weight.detach().contiguous() for l in m.linears()], [l.bias.detach().contiguous() for l in m.linears()])


h, e, ge, ga, aggr, gh = rows(N, 5), rows(E, 4), rows(E, 4), rows(N, 4), rows(N, 4), rows(N, 5)
Wr, br = params(14, 4)
Wo, bo = params(9, 5)
mr, mo = ops._fill_mlp(Wr, br), ops._fill_mlp(Wo, bo)


def relational():
    return B.mlp_backward_raw([h, h, e], [gi.tgt, gi.src, None], [True, True, True], Wr, br, n_rows=E,
                              epilogue=_capi.EPI_NONE, ca=0.0, cb=1.0, gout=[(ge, None), (ga, gi.tgt)],
                              need_seg=[True, True, True], want_dw=True, mlp=mr, gidx=[None, gi.spos_inv, None])


def object_model():
    return B.mlp_backward_raw([h, aggr], [None, None], [True, False], Wo, bo, n_rows=N, epilogue=_capi.EPI_RESIDUAL,
                              ca=0.7071, cb=0.7071, gout=[(gh, None)], need_seg=[True, True], want_dw=True, mlp=mo)


es = [rows(E, 4) for _ in range(4)]
Wh, bh = params(26, 1)
mh = ops._fill_mlp(Wh, bh)
gw = torch.randn(E, 1, device=dev)


def head():
    return B.mlp_backward_raw([h, h] + es, [gi.src, gi.tgt, None, None, None, None], [False] * 6, Wh, bh, n_rows=E,
                              epilogue=_capi.EPI_SIGMOID, ca=0.001, cb=0.998, gout=[(gw, None)], need_seg=[True] * 6,
                              want_dw=True, mlp=mh, gidx=[gi.spos_inv] + [None] * 5)


def timed(fn, flags):
    B._DEBUG_FLAGS = flags
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t), out


def flat(out):
    sl, gW, gb = out
    return [x for x in list(sl) + list(gW) + list(gb) if x is not None]


for name, fn, nrows in (("relational", relational, E), ("head", head, E), ("object", object_model, N)):
    forms = (0, 128) if args.only is None else (args.only,)
    ref = None
    for f in forms:
        _, out = timed(fn, f)
        if ref is None:
            ref = flat(out)
        else:
            same = all(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
                       for a, b in zip(ref, flat(out)))
            print(f"{name}: buffer form == generic form bit for bit: {same}")
    ts = {f: [] for f in forms}
    for _ in range(args.iters):
        for f in forms:
            t, out = timed(fn, f)
            del out
            ts[f].append(t)
    for f in forms:
        v = sorted(ts[f])
        print(f"{name:10s} {'buffer I/O ' if f == 0 else 'generic I/O'}: median {v[len(v) // 2]:.3f} ms  min {v[0]:.3f} ms  ({nrows} rows)")
B._DEBUG_FLAGS = 0
