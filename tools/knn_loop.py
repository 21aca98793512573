#!/usr/bin/env python3
"""A few cfg5 neighbour searches in a row (for rocprofv3 --kernel-trace): 200 k hits, k, r from argv."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_tracking_amd import ops
from gnn_tracking_amd.synthetic import make_pileup_cloud

k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
r = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
x = make_pileup_cloud(500, 200_000).cuda()
for _ in range(6):
    ei = ops.knn_graph(x, k, r if r > 0 else None)
torch.cuda.synchronize()
print(ei.shape)
