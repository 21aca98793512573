#!/usr/bin/env python3
"""Headline benchmark: edges/s (forward + BCE + backward + grad all-reduce + Adam) of the
ECForGraphTCN edge classifier on synthetic TrackML-shaped hit graphs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2]/[3]): per GPU a batch of 32
events x (150 000 hits, 2 000 000 edges) collated into one disjoint graph (N = 4.8 M,
E = 64 M), model ECForGraphTCN(node_indim=14, edge_indim=4, L_ec=3, hidden_dim=40), fp32.
Weak scaling: every rank owns its own 32 events; the only collective is one RCCL
all-reduce of the flat gradient buffer per step.  A step does everything a training
step does, including rebuilding the graph index from the raw COO edge_index (the cache
is cleared every step: a new batch would arrive every step in real training).

Prints ONE JSON line (rank 0).  ``roofline`` describes the dominant kernel (the fused
gather-MLP backward, fp32 MFMA bound at hidden width 40) from HIP-event timings taken
inside the timed region; ``cpu_baseline`` is the CPU oracle (oracle/ref_cpu.py, a
restatement of the reference pinned against it) timed on this box's host cores on one
event of the same workload.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import dist as gdist  # noqa: E402
from gnn_tracking_amd import ops, synthetic  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix = fp32 vector peak
PEAK_HBM_TBPS = 8.0            # MI355X_MICROARCH.md: HBM3E spec peak
TRAFFIC_PROFILES = {"f32": "r01_hbm_traffic_v5.json", "bf16": "r01_hbm_traffic_bf16_v8.json"}

WORKLOADS = {
    # name: (events per GPU, hits per event, edges per event, model kwargs)
    "cfg3": (32, 150_000, 2_000_000, dict(L_ec=3, hidden_dim=40)),
    "cfg2": (1, 10_000, 100_000, dict(L_ec=3, hidden_dim=40)),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--events", type=int, default=None, help="override events per GPU")
    ap.add_argument("--dtype", default="bf16", choices=("bf16", "f32"),
                    help="storage / MFMA-input type of the activations (cfg3 names bf16 storage, "
                         "fp32 accumulate); parameters and their gradients are fp32 in both")
    ap.add_argument("--index", default="inline", choices=("inline", "prefetch"),
                    help="graph index built inline in the step (default), or for the next batch on a side stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(event, model, iters: int) -> dict:
    """Oracle fwd+BCE+bwd on ONE event on the host cores (rank 0, N=1 only).

    torch's CPU kernels on the tiny MLP widths of this model stop scaling long before
    the core count of a GPU host, so the thread count is probed first on a 1/8-event
    slice and the fastest one is used (and reported as ``cores``)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as O

    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x, ei, ea, y = event.x.cpu(), event.edge_index.cpu(), event.edge_attr.cpu(), event.y.cpu()
    hp = model.hparams

    def one(ei_, ea_, y_):
        ps = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        out = O.ec_for_graph_tcn(x, ei_, ea_, ps, L_ec=hp.L_ec, alpha=hp.alpha)
        loss = O.edge_weight_bce_loss(out["W"], y_.float())
        torch.autograd.grad(loss, list(ps.values()))

    ncpu = os.cpu_count() or 1
    E = int(ei.shape[1])
    sl = slice(0, max(E // 8, 1))
    probe = {}
    for th in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        one(ei[:, sl], ea[sl], y[sl])
        t0 = time.perf_counter()
        one(ei[:, sl], ea[sl], y[sl])
        probe[th] = time.perf_counter() - t0
        if probe[th] > 1.4 * min(probe.values()):
            break  # past the optimum: more threads only get slower (256 threads take 45 s here)
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    one(ei, ea, y)  # warm-up (untimed)
    t0 = time.perf_counter()
    for _ in range(iters):
        one(ei, ea, y)
    dt = (time.perf_counter() - t0) / iters
    return {"value": E / dt, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"1 event ({x.shape[0]} hits, {E} edges) of the workload, "
                      f"{iters} timed fwd+BCE+bwd iterations after 1 warm-up, "
                      f"torch {torch.__version__} CPU, {threads} threads (fastest of "
                      f"{sorted(probe)} probed on a 1/8-event slice, stopping once slower; host has {ncpu} CPUs)",
            "s_per_iter": dt, "thread_probe_s": probe}


def measured_traffic(kernel: str, rows_per_launch: float, dtype: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE and
    WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes; see the JSON's _about),
    scaled by rows when this run's launch size differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", TRAFFIC_PROFILES[dtype])
    if not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f)["kernels"].get(kernel)
    if rec is None:
        return None
    return rec["hbm_bytes_per_launch"] * rows_per_launch / rec["rows_per_launch"]


def main():
    args = parse()
    rank, local, world = gdist.init_process_group_from_env()
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the package has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n_ev, n_hits, n_edges, mkw = WORKLOADS[args.workload]
    if args.events:
        n_ev = args.events
    torch.manual_seed(0)  # identical initial weights on every rank
    model = G.ECForGraphTCN(node_indim=14, edge_indim=4, **mkw).to(dev)
    flat = gdist.FlatParameters(model)
    opt = torch.optim.Adam([flat.flat_param], lr=1e-4, weight_decay=1e-4)
    loss_fct = G.EdgeWeightBCELoss()

    events = [synthetic.make_event(100 + rank * n_ev + i, n_hits, n_edges, dev)
              for i in range(n_ev)]
    batch = G.collate(events)
    first_event_cpu = events[0].cpu() if (rank == 0 and world == 1) else None
    del events
    yf = batch.y.float()
    E_local = batch.num_edges

    # Every step pays one graph-index build.  --index inline (default): at the start of the forward.
    # --index prefetch: the build for the NEXT batch runs on the loader's side stream
    # while this step computes - the index depends on the input edge list only, and
    # io.PrefetchLoader(build_index=True) does exactly this one batch ahead - so a step
    # consumes the index the previous step built.  Two batch objects (own edge_index tensors,
    # same content) alternate so that "next batch" is a different tensor, as with a loader.
    side = torch.cuda.Stream(dev) if args.index == "prefetch" else None
    batches = [batch]
    if side is not None:
        import copy
        other = copy.copy(batch)
        other.edge_index = batch.edge_index.clone()
        batches.append(other)
    counter = [0]

    def step():
        cur = batches[counter[0] % len(batches)]
        nxt = batches[(counter[0] + 1) % len(batches)]
        counter[0] += 1
        if side is None:
            ops.clear_graph_index_cache()
        else:
            ops.prefetch_graph_index(nxt.edge_index, nxt.num_nodes, side)
        flat.zero_grad()
        with G.bf16_storage(args.dtype == "bf16"):
            out = model(cur)
            loss = loss_fct(w=out["W"], y=yf, edge_index=cur.edge_index, pt=cur.pt)
            loss.backward()
        flat.all_reduce_grads()
        opt.step()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.item())

    if rank == 0:
        ks = timer.summary()
        dom = max(ks, key=lambda k: ks[k]["ms"]) if ks else None
        roof = None
        kernels = {}
        for k, d in ks.items():
            kernels[k] = {"launches": d["launches"], "avg_ms": d["ms"] / d["launches"],
                          "tflops": d["flops"] / (d["ms"] * 1e-3) / 1e12,
                          "alg_GBps": d["bytes"] / (d["ms"] * 1e-3) / 1e9}
            # HBM bytes actually moved (committed PMC passes) over THIS run's launch time
            tr = measured_traffic(k, d["rows"] / d["launches"], args.dtype)
            if tr is not None:
                gbps = tr / (d["ms"] / d["launches"] * 1e-3) / 1e9
                kernels[k].update({"traffic": tr, "hbm_GBps": gbps, "hbm_frac": gbps / (PEAK_HBM_TBPS * 1e3)})
        if dom:
            d = ks[dom]
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
            gbps = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            common = {"kernel": dom, "launches": d["launches"], "avg_launch_ms": d["ms"] / d["launches"],
                      "alg_flops_per_launch": d["flops"] / d["launches"],
                      "alg_bytes_per_launch": d["bytes"] / d["launches"],
                      "traffic": measured_traffic(dom, d["rows"] / d["launches"], args.dtype)}
            if args.dtype == "bf16":
                # bf16 MFMA (2.5 PFLOP/s) leaves the fused kernels HBM / issue bound: the
                # roofline that bounds them is HBM bandwidth
                roof = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_TBPS * 1e3, "unit": "GB/s",
                        "frac": gbps / (PEAK_HBM_TBPS * 1e3), **common,
                        "mfma_tflops_algorithmic": tf}
            else:
                roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / PEAK_F32_MFMA_TFLOPS, **common,
                        "hbm_frac_algorithmic": gbps / (PEAK_HBM_TBPS * 1e3)}
        cpu = None
        if world == 1 and not args.no_cpu_baseline and first_event_cpu is not None:
            cpu = cpu_baseline(first_event_cpu, model, args.cpu_iters)
        total_edges = E_local * world * args.steps
        line = {
            "metric": "edges_per_sec_fwd_bwd",
            "value": total_edges / dt,
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: per GPU {n_ev} events x {n_hits} hits x {n_edges} "
                            f"edges collated (N={batch.num_nodes}, E={E_local}); "
                            f"ECForGraphTCN(node_indim=14, edge_indim=4, L_ec={mkw['L_ec']}, "
                            f"hidden_dim={mkw['hidden_dim']}); step = graph index "
                            + ("(built for the next batch on the loader's side stream during the step) "
                               if args.index == "prefetch" else "(inline) ")
                            + "+ forward + BCE + backward + grad all-reduce + Adam",
                "graph_index": args.index,
                "global_edges_per_step": E_local * world,
                "parallelism": f"dp{world} (events sharded, flat-gradient RCCL all-reduce)",
            },
            "edge_layers_per_sec": total_edges * mkw["L_ec"] / dt,
            "final_loss": final_loss,
            "roofline": roof,
            "kernels": kernels,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
