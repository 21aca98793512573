"""One full-size event per step (the reference's DataLoader default, batch_size = 1): a few eager steps for a kernel
trace.   rocprofv3 --kernel-trace -d /tmp/p_1 -o k -- python tools/one_event_step.py [--dtype bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import ops, synthetic, training  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--separate", action="store_true", help="45 separate parameter tensors instead of the bucket")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = G.ECForGraphTCN(node_indim=14, edge_indim=4, L_ec=3, hidden_dim=40).to(dev)
if args.separate:
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=True)
    mod = training.ECModule(model, loss_fct=G.EdgeWeightBCELoss(), bf16=args.dtype == "bf16", optimizer=lambda p: opt,
                            scheduler=None)
else:
    from gnn_tracking_amd import dist as gdist
    mod = training.ECModule(model, loss_fct=G.EdgeWeightBCELoss(), bf16=args.dtype == "bf16",
                            flat=gdist.FlatParameters(model), scheduler=None,
                            optimizer=lambda p: torch.optim.Adam(p, lr=1e-4, weight_decay=1e-4, capturable=True))
batch = G.collate([synthetic.make_event(100, 150_000, 2_000_000, dev)])
for _ in range(args.steps):
    ops.clear_graph_index_cache()
    mod.optimisation_step(batch)
torch.cuda.synchronize()
