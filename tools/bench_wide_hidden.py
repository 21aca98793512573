"""Hidden width 64 .. 95 in bf16 storage: the five / six-hidden-tile instantiations of the fused kernels
against the library-GEMM path (`ops._wide_mlp`) the same model took before, one training step of
ECForGraphTCN on collated cfg3-sized events.

    python tools/bench_wide_hidden.py [--hidden 64] [--events 8] [--steps 5]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import dist as gdist, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hidden", type=int, default=64)
ap.add_argument("--events", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--mode", default="all", choices=("all", "fused", "library", "fp32"))
args = ap.parse_args()
dev = torch.device("cuda", 0)
batch = G.collate([synthetic.make_event(100 + i, 150_000, 2_000_000, dev) for i in range(args.events)])
E = int(batch.edge_index.shape[1])
loss_fct = G.EdgeWeightBCELoss()


def run(tag, bf16, fused_rule=None):
    torch.manual_seed(0)
    model = G.ECForGraphTCN(node_indim=14, edge_indim=4, L_ec=3, hidden_dim=args.hidden).to(dev)
    flat = gdist.FlatParameters(model)
    opt = torch.optim.Adam([flat.flat_param], lr=1e-4)
    orig = ops._fused_supported
    if fused_rule is not None:
        ops._fused_supported = fused_rule

    def step():
        ops.clear_graph_index_cache()
        flat.zero_grad()
        with G.bf16_storage(bf16):
            out = model(batch)
            loss = loss_fct(w=out["W"], y=batch.y, edge_index=batch.edge_index, pt=batch.pt)
            loss.backward()
        opt.step()
        return loss

    try:
        for _ in range(2):
            loss = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
    finally:
        ops._fused_supported = orig
    print(f"{tag:34s}: {ms:8.2f} ms/step  {E / ms / 1e6:7.3f} G edges/s  loss {float(loss.detach()):.6f}", flush=True)


def old_rule(segs, weights, biases, bf16):   # the limits before the five / six-tile instantiations
    ok = ops_fused(segs, weights, biases, bf16)
    if bf16 and ok:
        hidden = int(weights[0].shape[0])
        ok = hidden + (1 if any(b is not None for b in biases) else 0) <= 64
    return ok


ops_fused = ops._fused_supported
print(f"ECForGraphTCN(hidden_dim={args.hidden}), {args.events} events, E = {E}")
if args.mode in ("all", "fused"):
    run("bf16 storage, fused kernels", True)
if args.mode in ("all", "library"):
    run("bf16 storage, library GEMMs (before)", True, old_rule)
if args.hidden <= 64 and args.mode in ("all", "fp32"):
    run("fp32, fused kernels", False)
