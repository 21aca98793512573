"""Pruned against brute-force neighbour search on a 12-dimensional cloud (200 k hits): the 9..16-dimensional path of
the sorted-chunk search (four bits per dimension in the sort key, at most four queries per wave)."""
import sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_tracking_amd import ops
from gnn_tracking_amd.synthetic import make_pileup_cloud
x = make_pileup_cloud(500, 200000, 12).cuda()
for k, r in ((16, 1.0), (256, 1.0)):
    res = {}
    for flags in (2, 0):
        ops._KNN_FLAGS = flags
        ops.knn_graph(x[:9000].contiguous(), k, r)
        torch.cuda.synchronize(); t0 = time.perf_counter(); ei = ops.knn_graph(x, k, r); torch.cuda.synchronize()
        res[flags] = (time.perf_counter() - t0, ei)
    print(f"12-d k={k}: pruned {res[0][0]*1e3:.2f} ms | brute {res[2][0]*1e3:.1f} ms | identical {torch.equal(res[0][1], res[2][1])} edges {res[0][1].shape[1]}")
