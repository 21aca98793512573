"""Metric-learning stage as one training step on a 200 k-hit pile-up-like event:
GraphConstructionFCNN (14 -> 8-d embedding) forward, GraphConstructionHingeEmbeddingLoss (radius graph
at the reference's default max_num_neighbors = 256, r = 1) and backward; torch.profiler table."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import synthetic  # noqa: E402
from gnn_tracking_amd.losses_ml import GraphConstructionHingeEmbeddingLoss  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dev = torch.device("cuda", 0)
ev = synthetic.make_pileup_event(500, n, 8)
g = torch.Generator().manual_seed(0)
x = torch.cat([ev["x"], torch.rand(n, 6, generator=g) * 1.5], dim=1).to(dev)
pid = ev["particle_id"].to(dev)
# true edges: consecutive hits of the same particle (in index order)
order = torch.argsort(pid, stable=True)
same = (pid[order][1:] == pid[order][:-1]) & (pid[order][1:] > 0)
true_edges = torch.stack([order[:-1][same], order[1:][same]])
torch.manual_seed(0)
model = G.GraphConstructionFCNN(in_dim=14, hidden_dim=64, out_dim=8, depth=3).to(dev)
loss_fct = GraphConstructionHingeEmbeddingLoss()
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
data = G.Data(x=x, edge_index=true_edges, particle_id=pid, pt=ev["pt"].to(dev), eta=ev["eta"].to(dev),
              reconstructable=ev["reconstructable"].to(dev), batch=torch.zeros(n, dtype=torch.long, device=dev))


def step():
    opt.zero_grad(set_to_none=False)
    out = model(data)
    # (start near the input slice so that the radius graph has the density of config 5)
    h = 0.5 * out["H"] + data.x[:, :8]
    ret = loss_fct(x=h, particle_id=data.particle_id, batch=data.batch, true_edge_index=true_edges, pt=data.pt,
                   eta=data.eta, reconstructable=data.reconstructable)
    ret.loss.backward()
    opt.step()
    return ret


for _ in range(3):
    r = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    r = step()
torch.cuda.synchronize()
print(f"ML step n={n}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms; rep edges {int(r.extra_metrics['n_edges_rep'])}, "
      f"att edges {int(r.extra_metrics['n_edges_att'])}")
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=14, max_name_column_width=60))
