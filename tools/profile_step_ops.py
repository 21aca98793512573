#!/usr/bin/env python3
"""Which ATen ops (outside the gnntrk_* C-ABI calls) does one cfg3 training step launch?
torch.profiler table of one step, sorted by device time - finds stray conversions / copies.

    python tools/profile_step_ops.py [--events 8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import dist as gdist, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=8)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = G.ECForGraphTCN(node_indim=14, edge_indim=4, L_ec=3, hidden_dim=40).to(dev)
flat = gdist.FlatParameters(model)
opt = torch.optim.Adam([flat.flat_param], lr=1e-4, weight_decay=1e-4)
loss_fct = G.EdgeWeightBCELoss()
batch = G.collate([synthetic.make_event(100 + i, 150_000, 2_000_000, dev) for i in range(args.events)])
yf = batch.y   # (the dataset's bool labels, as bench.py passes them)


def step():
    ops.clear_graph_index_cache()
    flat.zero_grad()
    with G.bf16_storage(True):
        out = model(batch)
        loss = loss_fct(w=out["W"], y=yf, edge_index=batch.edge_index, pt=batch.pt)
        loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45,
                                                        max_name_column_width=48, max_shapes_column_width=60))
