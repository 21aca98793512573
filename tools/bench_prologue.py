"""Fixed cost per launch of the bf16 fused-MLP kernels: the relational shape forward / backward at row counts from one
tile up (HIP events around single launches, median).  time(rows -> 0) is the prologue (segment staging, slot plan,
weight-fragment packing) + launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import _capi, ops, ops_bf16 as B  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = G.MLP(14, 4, 40, L=3).to(dev)
W = [l.weight.detach().contiguous() for l in m.linears()]
bs = [l.bias.detach().contiguous() for l in m.linears()]
mlp = ops._fill_mlp(W, bs)


def rows(n, d):
    t = B.empty_rows(n, d, dev, zero=True)
    t.copy_(torch.randn(n, d, device=dev))
    return t


def med(fn, n=30):
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[n // 2]


for E in (16, 4096, 150_000, 2_000_000):
    N = max(E // 13, 2)
    h, e, ge = rows(N, 5), rows(E, 4), rows(E, 4)
    tgt = torch.randint(0, N, (E,), device=dev, dtype=torch.int32).sort().values
    src = torch.randint(0, N, (E,), device=dev, dtype=torch.int32)
    ident = torch.arange(E, device=dev, dtype=torch.int32)
    fwd = lambda: B.mlp_forward_raw([h, h, e], [tgt, src, None], [True, True, True], W, bs, n_rows=E, epilogue=_capi.EPI_NONE,
                                    ca=0.0, cb=1.0, res=None, out_idx=None, out_rows=E, mlp=mlp)
    bwd = lambda: B.mlp_backward_raw([h, h, e], [tgt, src, None], [True, True, True], W, bs, n_rows=E, epilogue=_capi.EPI_NONE,
                                     ca=0.0, cb=1.0, gout=[(ge, None)], need_seg=[True, True, True], want_dw=True, mlp=mlp,
                                     gidx=[None, ident, None])
    fwd(); bwd()
    print(f"rows {E:8d}: forward {med(fwd):7.1f} us   backward (+ partial reduction) {med(bwd):7.1f} us")
