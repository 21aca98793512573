"""GraphConstructionResIN at the reference's default width (hidden_dim = 40: 120 -> 40 -> 40 -> 40 relational
model, 80 -> 40 -> 40 -> 40 object model, 40-wide encoder outputs) in bf16 storage: the output-tile / wide-input
instantiations of the fused kernels against the library-GEMM path the model took before, forward + backward.

    python tools/bench_gc_resin.py [--hits 200000] [--edges 3000000] [--steps 5]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hits", type=int, default=200_000)
ap.add_argument("--edges", type=int, default=3_000_000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--mode", default="all", choices=("all", "fused", "library", "fp32", "fp32w"))
args = ap.parse_args()
dev = torch.device("cuda", 0)
ev = synthetic.make_event(7, args.hits, args.edges, dev)
data = G.Data(x=ev.x, edge_index=ev.edge_index, edge_attr=ev.edge_attr)
E = int(ev.edge_index.shape[1])
orig = ops._fused_supported


def old_rule(segs, weights, biases, bf16, epilogue=None):   # the limits before the output-tile kernels
    if bf16 and (int(weights[-1].shape[0]) > 16 or sum((int(s.t.shape[1]) + 3) // 4 for s in segs) > 15):
        return False
    return orig(segs, weights, biases, bf16, epilogue)


def run(tag, bf16, rule=None):
    torch.manual_seed(0)
    model = G.GraphConstructionResIN(node_indim=14, edge_indim=4, n_layers=1).to(dev)
    ops._fused_supported = rule or orig
    try:
        def step():
            model.zero_grad()
            with G.bf16_storage(bf16):
                h = model(data)["H"]
                h.float().square().mean().backward()
            return h
        for _ in range(2):
            h = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            h = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
    finally:
        ops._fused_supported = orig
    print(f"{tag:38s}: {ms:8.2f} ms fwd+bwd  {E / ms / 1e6:7.3f} G edges/s  |H| {float(h.float().abs().mean()):.5f}", flush=True)


print(f"GraphConstructionResIN(hidden_dim=40), {args.hits} hits, E = {E}")
if args.mode in ("all", "fused"):
    run("bf16 storage, fused kernels", True)
if args.mode in ("all", "library"):
    run("bf16 storage, library GEMMs (before)", True, old_rule)
if args.mode in ("all", "fp32", "fp32w"):
    run("fp32, wide fused kernels (mlp_wide)", False)
if args.mode in ("all", "fp32"):
    ops._WIDE_KERNEL = False
    try:
        run("fp32, library GEMMs (before round 4)", False)
    finally:
        ops._WIDE_KERNEL = True
