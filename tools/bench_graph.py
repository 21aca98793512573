"""cfg2 (one 10 k-hit event, 100 k edges: cache resident, launch bound) as a HIP graph:
the whole training step - graph index, forward, BCE, backward, Adam - is captured once
with torch.cuda.CUDAGraph (hipGraph underneath) and replayed.  Every gnntrk_* entry point
is stream-ordered, allocation-free and sync-free, which is what makes it capturable."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hits", type=int, default=10_000)
ap.add_argument("--edges", type=int, default=100_000)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--dtype", default="bf16", choices=("bf16", "f32"))
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = G.ECForGraphTCN(node_indim=14, edge_indim=4, L_ec=3, hidden_dim=40).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=True)
loss_fct = G.EdgeWeightBCELoss()
batch = G.collate([synthetic.make_event(1, args.hits, args.edges, dev)])
yf = batch.y.float()


def step():
    ops.clear_graph_index_cache()
    opt.zero_grad(set_to_none=False)
    with G.bf16_storage(args.dtype == "bf16"):
        out = model(batch)
        loss = loss_fct(w=out["W"], y=yf, edge_index=batch.edge_index, pt=batch.pt)
        loss.backward()
    opt.step()
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    step()
eager = timed(step, args.steps)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
g.replay()
torch.cuda.synchronize()
l0 = float(static_loss)
graph = timed(g.replay, args.steps)
print(f"cfg2 ({args.hits} hits, {args.edges} edges, {args.dtype}): eager {eager:.3f} ms/step "
      f"({args.edges / eager / 1e3:.1f} M edges/s) | hipGraph replay {graph:.3f} ms/step "
      f"({args.edges / graph / 1e3:.1f} M edges/s) | loss {l0:.5f} -> {float(static_loss.detach()):.5f}")
