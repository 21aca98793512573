"""Node order + graph index of the cfg3 batch (32 events x 150 000 hits x 2 000 000 edges), eager, HIP-event timed:
   GNNTRK_LIB=tools/_bin/variants/<name>/libgnntrk.so python tools/bench_remap.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import ops, synthetic  # noqa: E402

dev = torch.device("cuda", 0)
b = G.collate([synthetic.make_event(100 + i, 150_000, 2_000_000, dev) for i in range(32)])
ts = []
for i in range(12):
    ops.clear_graph_index_cache()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gi = ops.graph_index(b.edge_index, b.num_nodes, cache=False, carry_label=b.y, carry_rows=b.edge_attr,
                         order_by=(b.x, 1, b.batch, 32))
    e1.record()
    torch.cuda.synchronize()
    if i >= 2:
        ts.append(e0.elapsed_time(e1))
    del gi
print(os.environ.get("GNNTRK_LIB", "base").split("/")[-2] if "GNNTRK_LIB" in os.environ else "base",
      "index build ms: median %.3f  min %.3f" % (sorted(ts)[len(ts) // 2], min(ts)))
