"""The graph-index build (gnntrk_graph_index_build_ex) at the cfg3 size: own two-level counting sort
against the library radix-sort form, same arrays (checked), alternating, HIP-event times.

    python tools/bench_graph_index.py [--events 32] [--hits 150000] [--edges 2000000] [--reps 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=32)
ap.add_argument("--hits", type=int, default=150_000)
ap.add_argument("--edges", type=int, default=2_000_000)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--single", action="store_true", help="one giant unsorted graph of the same size instead")
ap.add_argument("--only", type=int, default=None, help="run only this form (for rocprofv3)")
args = ap.parse_args()
dev = torch.device("cuda", 0)

if args.single:
    N, E = args.events * args.hits, args.events * args.edges
    ei = torch.randint(0, N, (2, E), device=dev)
else:
    batch = G.collate([synthetic.make_event(100 + i, args.hits, args.edges, dev) for i in range(args.events)])
    ei, N = batch.edge_index, batch.num_nodes
    E = int(ei.shape[1])
print(f"N = {N}, E = {E}, {'single graph' if args.single else f'{args.events} collated events'}")


def timed(flags):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gi = ops.graph_index(ei, N, cache=False, flags=flags)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), gi


forms = {0: "own counting sort", 1: "library radix sort"}
if args.only is not None:
    forms = {args.only: forms[args.only]}
ref = None
for f in forms:
    _, gi = timed(f)
    arrs = [gi.perm, gi.tgt, gi.src, gi.rowptr_t, gi.rowptr_s, gi.spos, gi.spos_inv]
    if ref is None:
        ref = arrs
    else:
        assert all(torch.equal(a, b) for a, b in zip(ref, arrs)), "the two forms disagree"
        print("arrays identical")
    del gi
times = {f: [] for f in forms}
for _ in range(args.reps):
    for f in forms:
        t, gi = timed(f)
        del gi
        times[f].append(t)
for f, name in forms.items():
    ts = sorted(times[f])
    print(f"{name:20s}: median {ts[len(ts) // 2]:.3f} ms, min {ts[0]:.3f} ms  ({E / ts[len(ts) // 2] / 1e6:.1f} G edges/s)")
