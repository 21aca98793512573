#!/usr/bin/env python3
"""Headline benchmark: edges/s (forward + loss + backward + grad all-reduce + Adam) of the
ECForGraphTCN edge classifier on synthetic TrackML-shaped hit graphs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Without torchrun's environment ``--gpus N`` (N > 1) makes this script launch its own N ranks
(one process per GPU, RCCL); either way the line is only printed when the process group
really has N ranks (``ranks``, with the collective ``backend``; ``rccl_ranks`` only under nccl = RCCL).

Workloads (BASELINE.json ``configs``):

* ``cfg3`` (default; configs[2]): per GPU a batch of 32 events x (150 000 hits, 2 000 000
  edges) collated into one disjoint graph (N = 4.8 M, E = 64 M), ``ECForGraphTCN(14, 4,
  L_ec=3, hidden_dim=40)``, bf16 storage / fp32 accumulate (``--dtype f32``: reference
  precision).  Weak scaling: every rank owns its own 32 events.
* ``cfg4`` (configs[3]): 256 events with N ~ U(100 k, 200 k) hits (E = 40/3 N), greedy
  size-balanced into 8 shards of 32 events; the SAME 256 events at every rank count (strong
  scaling): a rank owns 8/N shards and runs them as micro-batches of one optimisation step.
* ``cfg2`` (configs[1]): one 10 000-hit / 100 000-edge event (cache resident, launch bound).
* ``cfg5`` (configs[4]): the object-condensation step on a 200 000-hit pile-up event: kNN
  graph build (MLGraphConstruction) -> GraphTCN -> CondensationLossRG -> backward -> Adam.

A step does everything a training step does, including rebuilding the graph index from the
raw COO edge_index (the cache is cleared every step: a new batch would arrive every step).
The only collective is one RCCL all-reduce of the flat gradient buffer per step.

Prints ONE JSON line of at most 6 KB on stdout (rank 0).  ``roofline`` describes the dominant fused
gather-MLP kernel from HIP-event timings taken inside the timed region; ``cpu_baseline`` is the CPU
oracle (oracle/ref_cpu.py, a restatement of the reference pinned against it) timed on this box's
host cores on one event of the same workload; ``kernels`` / ``extra`` are digests (a few numbers per
kernel / per short run of the other configurations at N = 1): the full records go to
``bench_extra.json`` next to this file (and under ``gpurun_out/``) and to stderr (``emit``).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import gnn_tracking_amd as G  # noqa: E402
from gnn_tracking_amd import dist as gdist  # noqa: E402
from gnn_tracking_amd import ops, synthetic, training  # noqa: E402
from gnn_tracking_amd.precision import bf16_storage  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix = fp32 vector peak
PEAK_HBM_TBPS = 8.0            # MI355X_MICROARCH.md: HBM3E spec peak
TRAFFIC_PROFILES = {"f32": "r01_hbm_traffic_v5.json", "bf16": "r06_hbm_traffic_bf16.json"}
TRAFFIC_FALLBACK = {"bf16": "r05_hbm_traffic_bf16.json"}
PIPE_PROFILES = {"cfg5": "r06_pipe_util_cfg5.json", "dbscan": "r06_pipe_util_dbscan.json"}
PIPE_FALLBACK = {"cfg5": "r05_pipe_util_cfg5.json", "dbscan": "r05_pipe_util_dbscan.json"}

TRAFFIC_NOTES: dict = {}   # kernel -> which committed profile its traffic figure came from, and whether it is stale

EC_MODEL = dict(L_ec=3, hidden_dim=40)
CFG4_EVENTS, CFG4_SHARDS = 256, 8


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=("cfg2", "cfg3", "cfg4", "cfg5"))
    ap.add_argument("--events", type=int, default=None, help="override the event count (cfg3: per GPU; cfg4: total)")
    ap.add_argument("--dtype", default=None, choices=("bf16", "f32"),
                    help="storage / MFMA-input type of the activations (cfg3 names bf16 storage, "
                         "fp32 accumulate); parameters and their gradients are fp32 in both")
    ap.add_argument("--index", default="inline", choices=("inline", "prefetch", "resident"),
                    help="graph index built inline in every step (default: as if a new batch arrived every "
                         "step), for the next batch on a side stream, or kept resident per batch (what epochs "
                         ">= 2 over a dataset held in HBM see: 28 B/edge of index next to 42 B/edge of inputs)")
    ap.add_argument("--node-ids", default="random", choices=("random", "phi"),
                    help="synthetic events with hit ids random w.r.t. the geometry (default: worst case for the "
                         "node-row gathers) or numbered by phi (edges join neighbouring ids: best case)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other configurations")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--backend", default=None, choices=("nccl", "gloo"),
                    help="process-group backend (default: nccl = RCCL).  gloo also works with device tensors "
                         "(through the host) and lets tests run two ranks on ONE GPU")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY: replace the HIP workload by a CPU toy model (gloo) to exercise the "
                         "launcher / barrier / reduction control flow; the line is marked as a stub")
    a = ap.parse_args(argv)
    if a.dtype is None:   # cfg3 / cfg4 name bf16 storage; cfg5's parity bar (1e-5) is an fp32 one
        a.dtype = "f32" if a.workload == "cfg5" else "bf16"
    return a


# ------------------------------------------------------------------------------- launcher
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str]) -> int:
    """``python bench.py --gpus N`` without torchrun: start N copies of this script, one per
    GPU, with torchrun's environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Rank 0 inherits stdout (it prints the line).  Returns the worst exit code."""
    if "--stub" not in argv:
        from gnn_tracking_amd import _build
        if _build.have_hipcc():
            _build.build_lib()   # (stamp check; the ranks then all find the library up to date)
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   GNNTRK_BENCH_SELF_LAUNCHED="1")
        out = None if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, stdout=out))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    for q in pending:       # one rank failed: the others would wait in a collective
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# --------------------------------------------------------------------------- CPU baseline
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
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        one(ei, ea, y)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": E / dt, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"1 event ({x.shape[0]} hits, {E} edges) of the workload, median of "
                      f"{iters} timed fwd+BCE+bwd iterations after 1 warm-up, "
                      f"torch {torch.__version__} CPU, {threads} threads (fastest of "
                      f"{sorted(probe)} probed on a 1/8-event slice, stopping once slower; host has {ncpu} CPUs)",
            "s_per_iter": dt, "s_all_iters": times, "thread_probe_s": probe}


def parity_check(event_cpu, model, dtype: str, dev) -> dict:
    """The GPU's edge weights and loss on event 0 of the workload (the run's precision, the parameters the run ended
    with) against the CPU oracle on the same event: fp32 against ``oracle.ec_for_graph_tcn`` (bar 1e-5), bf16 storage
    against the oracle's restatement of the rounding contract ``ec_for_graph_tcn_bf16`` (bar 5e-4: rare one-ulp flips of
    a hidden activation, 1.1e-4 measured) with the distance to the fp32 oracle next to it.  The checker, not the thing measured."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as O

    hp = model.hparams
    p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    d = event_cpu.to(dev)
    was_training = model.training
    model.eval()
    with torch.no_grad(), G.bf16_storage(dtype == "bf16"):
        out = model(d)
        w = torch.as_tensor(out["W"]).float()
        loss = float(G.EdgeWeightBCELoss()(w=out["W"], y=d.y.float(), edge_index=d.edge_index, pt=d.pt))
    model.train(was_training)
    w = w.cpu()
    x, ei, ea, y = event_cpu.x, event_cpu.edge_index, event_cpu.edge_attr, event_cpu.y
    t0 = time.perf_counter()
    with torch.no_grad():
        ref32 = O.ec_for_graph_tcn(x, ei, ea, p, L_ec=hp.L_ec, alpha=hp.alpha)
        rloss32 = float(O.edge_weight_bce_loss(ref32["W"], y.float()))
        rec = {"event": f"event 0 of the workload ({x.shape[0]} hits, {ei.shape[1]} edges), parameters after the timed steps",
               "dtype": dtype, "max_abs_W_vs_fp32_oracle": float((w - ref32["W"]).abs().max()),
               "loss_gpu": loss, "loss_fp32_oracle": rloss32}
        if dtype == "bf16":
            ref16 = O.ec_for_graph_tcn_bf16(x, ei, ea, p, L_ec=hp.L_ec, alpha=hp.alpha)
            err, bound = float((w - ref16["W"]).abs().max()), 5e-4   # (1.1e-4 measured; tests/parity_cases.py: BF16_ORACLE_W)
            rec.update({"oracle": "oracle/ref_cpu.py: ec_for_graph_tcn_bf16 (the kernels' rounding contract restated)",
                        "max_abs_W": err, "bound": bound,
                        "loss_abs": abs(loss - float(O.edge_weight_bce_loss(ref16["W"], y.float()))), "loss_bound": 1e-4})
        else:
            err, bound = rec["max_abs_W_vs_fp32_oracle"], 1e-5
            rec.update({"oracle": "oracle/ref_cpu.py: ec_for_graph_tcn (fp32)", "max_abs_W": err, "bound": bound,
                        "loss_abs": abs(loss - rloss32), "loss_bound": 1e-5})
    rec["ok"] = bool(rec["max_abs_W"] <= rec["bound"] and rec["loss_abs"] <= rec["loss_bound"])
    rec["oracle_seconds"] = time.perf_counter() - t0
    return rec


def library_id() -> dict:
    """What the loaded libgnntrk.so is: sha256 of the file and of the kernel sources it was built from.  Committed
    profiles carry the source hash they were taken on (tools/make_*_json.py); a block read from a profile of another
    build says so (``stale``) instead of passing for a measurement of this one."""
    import hashlib
    from gnn_tracking_amd import _build
    h = hashlib.sha256()
    for f in sorted(list(_build.CSRC.glob("*.hip")) + list(_build.CSRC.glob("*.h")) + list(_build.CSRC.glob("*.inc"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    lib = _build.LIB
    return {"csrc_sha256": h.hexdigest()[:16],
            "lib_sha256": hashlib.sha256(lib.read_bytes()).hexdigest()[:16] if lib.exists() else None}


def _profile_note(rec_file: dict) -> dict:
    """``source`` fields of a block taken from a committed profile: which build it describes, and whether that is
    the build running now."""
    built = rec_file.get("_csrc_sha256")
    cur = library_id()["csrc_sha256"]
    return {"source": "committed_profile", "profile_csrc_sha256": built, "library_csrc_sha256": cur,
            "stale": built != cur}


def access_floor(wl, dev, iters: int = 6) -> dict:
    """Time of the I/O SKELETON of the dominant kernel (the relational backward with three upstream terms:
    ``mlp16_bwd_skel_kernel``, debug_flags & 4096 - every load and store of the real kernel, the same descriptors,
    prefetch distance and occupancy, no arithmetic) on THIS run's batch and node numbering, next to the real kernel
    launched the same way.  Says how much of the kernel's time its access pattern costs whatever the instruction
    stream does: the floor an instruction diet cannot go below, and what a better layout has to move."""
    from gnn_tracking_amd import _capi, locality, ops_bf16 as B

    b = wl.batches[0]
    col = locality.key_column(b.x)
    bt = getattr(b, "batch", None)
    gi = ops.graph_index(b.edge_index, b.num_nodes, order_by=None if col is None else (b.x, col, bt))
    N, E = b.num_nodes, gi.n_edges
    g = torch.Generator(device=dev).manual_seed(0)

    def rows(n, d):
        t = B.empty_rows(n, d, dev, zero=True)
        t.copy_(torch.randn(n, d, device=dev, generator=g))
        return t

    h, e, ge, gt, ga = rows(N, 5), rows(E, 4), rows(E, 4), rows(E, 4), rows(N, 4)
    m = G.MLP(14, 4, 40, L=3).to(dev)
    W = [l.weight.detach().contiguous() for l in m.linears()]
    bs = [l.bias.detach().contiguous() for l in m.linears()]
    mlp = ops._fill_mlp(W, bs)

    def launch():
        return B.mlp_backward_raw([h, h, e], [gi.tgt, gi.src, None], [True, True, True], W, bs, n_rows=E,
                                  epilogue=_capi.EPI_NONE, ca=0.0, cb=1.0, gout=[(ge, None), (ga, gi.tgt), (gt, None)],
                                  need_seg=[True, True, True], want_dw=True, mlp=mlp, gidx=[None, gi.spos_inv, None])

    def timed(flags):
        old, B._DEBUG_FLAGS = B._DEBUG_FLAGS, flags
        try:
            ts = []
            for i in range(iters + 1):
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                out = launch()   # (the name is rebound for the head's launch below)
                s1.record()
                torch.cuda.synchronize()
                del out
                if i:
                    ts.append(s0.elapsed_time(s1))
            return sorted(ts)[len(ts) // 2]
        finally:
            B._DEBUG_FLAGS = old

    real, skel = timed(0), timed(4096)
    alg = 92.0 * E
    # the edge-weight head's backward the same way (h[src] | h[tgt] | four edge embeddings, fp32 upstream gradient)
    head = None
    launch_rel = launch
    try:
        es = [rows(E, 4) for _ in range(4)]
        mh = G.MLP(26, 1, 40, L=3).to(dev)
        Wh = [l.weight.detach().contiguous() for l in mh.linears()]
        bh = [l.bias.detach().contiguous() for l in mh.linears()]
        mlph = ops._fill_mlp(Wh, bh)
        gw = torch.randn(E, 1, device=dev, generator=g)

        def launch_head():
            return B.mlp_backward_raw([h, h, *es], [gi.src, gi.tgt, None, None, None, None], [False] * 6, Wh, bh, n_rows=E,
                                      epilogue=_capi.EPI_SIGMOID, ca=0.001, cb=0.998, gout=[(gw, None)],
                                      need_seg=[True] * 6, want_dw=True, mlp=mlph,
                                      gidx=[gi.spos_inv, None, None, None, None, None])

        launch = launch_head
        hreal, hskel = timed(0), timed(4096)
        launch = launch_rel
        halg = (2 * 16 + 4 * 8 + 4 + 8 + 2 * 16 + 4 * 8 + 4) * E   # rows read + ids + upstream + gradient rows + permutation ids
        head = {"kernel": "mlp16_bwd_skel_kernel<1, 3, 2, true, true, 2, IoHeadT<false> >", "skeleton_ms": hskel,
                "kernel_ms_isolated": hreal, "frac": halg / (hskel * 1e-3) / 1e9 / (PEAK_HBM_TBPS * 1e3),
                "share_of_kernel": hskel / hreal}
    except Exception as e:   # (the relational floor survives a failing head launch)
        head = {"error": f"{type(e).__name__}: {e}"}
    # ... and the relational FORWARD (x[tgt] | x[src] | e -> e~): mlp16_fwd_skel_kernel
    fwd = None
    try:
        def launch_fwd():
            return B.mlp_forward_raw([h, h, e], [gi.tgt, gi.src, None], [True, True, True], W, bs, n_rows=E,
                                     epilogue=_capi.EPI_NONE, ca=0.0, cb=1.0, res=None, out_idx=None, out_rows=E, mlp=mlp)

        launch = launch_fwd
        freal, fskel = timed(0), timed(4096)
        launch = launch_rel
        falg = 44.0 * E   # (as the bench counts the forward: bf16 rows + int32 ids in, e~ out)
        fwd = {"kernel": "mlp16_fwd_skel_kernel<1, 3, true, false, 4, true>", "skeleton_ms": fskel, "kernel_ms_isolated": freal,
               "frac": falg / (fskel * 1e-3) / 1e9 / (PEAK_HBM_TBPS * 1e3), "share_of_kernel": fskel / freal}
    except Exception as e_:   # noqa: BLE001
        fwd = {"error": f"{type(e_).__name__}: {e_}"}
    return {"kernel": "mlp16_bwd_skel_kernel<1, 3, 2, true, false, 2, IoRelational<3, false> >", "head": head, "forward": fwd,
            "what": "loads + stores of the relational backward (three upstream terms) without its arithmetic, same "
                    "occupancy and prefetch distance, this run's batch; launched alone (HIP events around launch + "
                    "partial reduction), median of %d" % iters,
            "node_order": "renumbered by data.x[:, %d] per event" % col if col is not None else "as given",
            "skeleton_ms": skel, "kernel_ms_isolated": real, "rows": E,
            "skeleton_GBps_algorithmic": alg / (skel * 1e-3) / 1e9,
            "frac": alg / (skel * 1e-3) / 1e9 / (PEAK_HBM_TBPS * 1e3),
            "share_of_kernel": skel / real}


def measured_traffic(kernel: str, rows_per_launch: float, dtype: str):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE and
    WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes; see the JSON's _about),
    scaled by rows when this run's launch size differs from the profiled one."""
    for name in (TRAFFIC_PROFILES.get(dtype), TRAFFIC_FALLBACK.get(dtype)):
        if not name:
            continue
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            doc = json.load(f)
        rec = doc["kernels"].get(kernel)
        if rec is not None:
            TRAFFIC_NOTES[kernel] = dict(_profile_note(doc), profile=name)
            return rec["hbm_bytes_per_launch"] * rows_per_launch / rec["rows_per_launch"]
    return None


def measured_pipe(which: str, kernel_prefix: str):
    """Measured pipe utilisation of the kernel whose name starts with ``kernel_prefix`` from the
    committed SQ-counter pass (tools/make_pipe_json.py): what the pruned searches / spatial loss
    passes are priced with - they visit a data-dependent few per cent of the pairs, so a flop count
    divided by their time is not a roofline."""
    name = PIPE_PROFILES[which]
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        name = PIPE_FALLBACK[which]
        path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        doc = json.load(f)
    ks = doc["kernels"]
    for name, rec in ks.items():
        if name.startswith(kernel_prefix) and "valu_busy" in rec:
            return {**_profile_note(doc),   # (NOT measured in this run: a constant read from profiles/)
                    "profile": name,
                    "bound": "valu", "achieved": rec["valu_busy"], "peak": 1.0,
                    "unit": "share of the chip's vector-issue cycles (SQ_INSTS_VALU x 4 / (1024 SIMDs x cycles); "
                            f"rocprofv3 --pmc, profiles/{name})",
                    "frac": rec["valu_busy"], "kernel": name, "kernel_us_under_pmc": rec["avg_us_under_pmc"],
                    "issue_share_of_wave_cycles": rec.get("issue_share"), "wait_share": rec.get("wait_share"),
                    "stall_share": rec.get("stall_share")}
    return None


# ------------------------------------------------------------------------------ workloads
class Workload:
    """What a rank runs per step.  ``step()`` returns the loss tensor of the last micro-batch."""

    name = ""
    scaling = "weak"
    edges_per_step_global = 0      # edges ALL ranks process in one step
    describe = ""
    info: dict = {}

    def step(self):
        raise NotImplementedError


def cfg4_event_sizes(n_events: int) -> list[tuple[int, int]]:
    """(hits, edges) of the cfg4 events: N ~ U(100 k, 200 k), E/N = 40/3 as cfg3 (numpy
    generator: the same list on every rank and host)."""
    import numpy as np

    g = np.random.default_rng(4)
    hits = g.integers(100_000, 200_001, size=n_events)
    return [(int(h), int(h * 40 // 3) // 2 * 2) for h in hits]


class ECWorkload(Workload):
    """cfg2 / cfg3 / cfg4: ECForGraphTCN training step(s) on collated events."""

    def __init__(self, args, rank: int, world: int, dev, *, workload: str, dtype: str, index: str = "inline",
                 hidden_dim: int | None = None, loader_renumbered: bool = False, cached_index: bool = False,
                 resident_dataset: bool = False):
        self.name, self.dtype = workload, dtype
        torch.manual_seed(0)  # identical initial weights on every rank
        model_kw = dict(EC_MODEL, **({"hidden_dim": hidden_dim} if hidden_dim else {}))
        self.model = G.ECForGraphTCN(node_indim=14, edge_indim=4, **model_kw).to(dev)
        self.flat = gdist.FlatParameters(self.model)
        self.module = training.ECModule(
            self.model, loss_fct=G.EdgeWeightBCELoss(), flat=self.flat, bf16=dtype == "bf16", scheduler=None,
            optimizer=lambda p: torch.optim.Adam(p, lr=1e-4, weight_decay=1e-4))
        self.first_event_cpu = None
        if workload == "cfg4":
            n_events = args.events or CFG4_EVENTS
            if CFG4_SHARDS % world:
                raise SystemExit(f"cfg4 has {CFG4_SHARDS} shards: --gpus must divide it, got {world}")
            sizes = cfg4_event_sizes(n_events)
            shards = gdist.shard_events([e for _, e in sizes], CFG4_SHARDS)
            loads = [sum(sizes[i][1] for i in s) for s in shards]
            mine = [s for j, s in enumerate(shards) if j % world == rank]
            self.batches = [G.collate([synthetic.make_event(1000 + i, *sizes[i], dev) for i in s]) for s in mine]
            self.scaling = "strong"
            self.edges_per_step_global = sum(loads)
            rank_loads = [sum(loads[j] for j in range(CFG4_SHARDS) if j % world == r) for r in range(world)]
            self.info = {"events": n_events, "shards": CFG4_SHARDS, "micro_batches_per_rank": len(mine),
                         "shard_edges_max_over_mean": max(loads) / (sum(loads) / len(loads)),
                         "rank_edges_max_over_mean": max(rank_loads) / (sum(rank_loads) / world),
                         "rank_edges": rank_loads}
            self.describe = (f"cfg4: {n_events} events, hits ~ U(100k, 200k), E = 40/3 N, greedy size-balanced into "
                             f"{CFG4_SHARDS} shards; the same events at every rank count, {len(mine)} "
                             f"micro-batch(es) of ~32 events per rank and step (total E = {sum(loads)})")
        else:
            n_ev, n_hits, n_edges = {"cfg3": (32, 150_000, 2_000_000), "cfg2": (1, 10_000, 100_000)}[workload]
            if args.events:
                n_ev = args.events
            seed0 = 1 if workload == "cfg2" else 100 + rank * n_ev
            events = [synthetic.make_event(seed0 + i, n_hits, n_edges, dev,
                                           phi_sorted_ids=getattr(args, "node_ids", "random") == "phi")
                      for i in range(n_ev)]
            if rank == 0 and world == 1:
                self.first_event_cpu = events[0].cpu()
            if loader_renumbered:   # (what io.GraphDataset(renumber=True) does per graph, once)
                from gnn_tracking_amd import io as gio
                events = [gio.renumber_nodes(e) for e in events]
            self.parts = None
            if cached_index:   # per-event indices built ONCE (labels / edge features carried, node order per event)
                from gnn_tracking_amd import locality
                self.parts = []
                for e in events:
                    col = locality.key_column(e.x)
                    self.parts.append(ops.graph_index(e.edge_index, e.num_nodes, cache=False, carry_label=e.y,
                                                      carry_rows=e.edge_attr if dtype == "bf16" else None,
                                                      order_by=None if col is None else (e.x, col, None)))
            self.dataset = None
            if resident_dataset:   # the product's loader for static datasets: per-event indices kept, batches collated + placed
                from gnn_tracking_amd import io as gio
                self.dataset = gio.ResidentDataset(events, dev, bf16=dtype == "bf16")
                for i in range(len(events)):
                    self.dataset.event(i)
            self.batches = [G.collate(events)]
            del events
            b = self.batches[0]
            self.edges_per_step_global = b.num_edges * world
            self.describe = (f"{workload}: per GPU {n_ev} events x {n_hits} hits x {n_edges} edges collated "
                             f"(N={b.num_nodes}, E={b.num_edges})"
                             + ("; hit ids numbered by phi (best-case gather locality)"
                                if getattr(args, "node_ids", "random") == "phi" else ""))
        self.describe += (f"; ECForGraphTCN(node_indim=14, edge_indim=4, L_ec={EC_MODEL['L_ec']}, "
                          f"hidden_dim={EC_MODEL['hidden_dim']}); step = graph index "
                          + ("(built for the next batch on the loader's side stream during the step) "
                             if index == "prefetch" else "(RESIDENT: built once per batch, not in the timed steps) "
                             if index == "resident" else "(inline) ")
                          + "+ forward + BCE + backward + grad all-reduce + Adam")
        # --index prefetch: the build for the NEXT batch runs on the loader's side stream while
        # this step computes (io.PrefetchLoader(build_index=True) does exactly this one batch
        # ahead).  Two batch objects (own edge_index tensors, same content) alternate so that
        # "next batch" is a different tensor, as with a loader.
        self.resident = index == "resident"
        self.side = torch.cuda.Stream(dev) if index == "prefetch" else None
        if self.side is not None:
            if len(self.batches) != 1:
                raise SystemExit("--index prefetch is wired for the single-batch workloads")
            import copy
            other = copy.copy(self.batches[0])
            other.edge_index = self.batches[0].edge_index.clone()
            self.batches.append(other)
        self.counter = 0
        self.stage = _StageTimer()

    def step(self):
        opt = self.module.configure_optimizers()
        self.module.zero_grad()
        if self.side is not None:
            cur = self.batches[self.counter % 2]
            nxt = self.batches[(self.counter + 1) % 2]
            self.counter += 1
            from gnn_tracking_amd import locality
            ops.prefetch_graph_index(nxt.edge_index, nxt.num_nodes, self.side,
                                     x=None if locality.order_column(nxt) is None else nxt.x, batch=getattr(nxt, "batch", None))
            loss = self.module.backward_step(cur)
        elif getattr(self, "dataset", None) is not None:
            # one epoch of one batch: the events in a new order, collated on the device, their indices placed
            self.counter += 1
            self.batches = None
            for b in self.dataset.batches(len(self.dataset), shuffle=True, seed=self.counter):
                loss = self.module.backward_step(b)
            del b
        else:
            n = len(self.batches)
            for b in self.batches:
                if not self.resident:
                    ops.clear_graph_index_cache()   # a new batch every step: every step pays its index
                if getattr(self, "parts", None) is not None:   # ... as a copy of the cached per-event indices
                    ops.place_graph_indices(self.parts, b)
                loss = self.module.backward_step(b, scale=1.0 / n)
        with self.stage("allreduce_adam"):
            self.flat.all_reduce_grads()
            opt.step()
        return loss

    def stages(self) -> dict:
        return self.stage.summary()


class _StageTimer:
    """HIP-event brackets around the stages of a step (events on the launch stream; read after
    the timed region)."""

    def __init__(self):
        self.rec, self.on = [], False

    def __call__(self, name):
        timer = self

        class _Ctx:
            def __enter__(self):
                if timer.on:
                    self.e0 = ops.timing_event()
                    self.e1 = ops.timing_event()
                    self.e0.record()

            def __exit__(self, *exc):
                if timer.on:
                    self.e1.record()
                    timer.rec.append((name, self.e0, self.e1))
        return _Ctx()

    def summary(self) -> dict:
        out: dict = {}
        for name, e0, e1 in self.rec:
            d = out.setdefault(name, [0, 0.0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
        return {k: {"calls": n, "avg_ms": ms / n} for k, (n, ms) in out.items()}


class TCWorkload(Workload):
    """cfg5 (BASELINE.json configs[4]): the object-condensation training step of the reference's
    ``TCModule`` (training/tc.py:50-84) on one 200 000-hit pile-up-like event per GPU:
    ``MLGraphConstruction`` (HIP kNN in the 8-d latent slice, k = 16, r = 1; labels; edge
    features) -> ``GraphTCN`` -> ``CondensationLossRG`` -> backward -> Adam."""

    name, scaling, dtype = "cfg5", "weak", "f32"
    K_NN, DIM = 16, 8

    def __init__(self, args, rank: int, world: int, dev, dtype: str = "f32"):
        import numpy as np

        self.dtype = dtype
        n = args.events or 200_000     # --events overrides the hit count here
        ev = synthetic.make_pileup_event(500 + rank, n, self.DIM)
        g = np.random.default_rng(500 + rank)
        extra = torch.from_numpy(g.uniform(0, 1.5, size=(n, 6)).astype(np.float32))
        x = torch.cat([ev["x"], extra], dim=1)   # 14 node features; the first 8 are the latent coordinates
        self.data = G.Data(x=x.to(dev), edge_index=torch.zeros(2, 0, dtype=torch.long, device=dev),
                           particle_id=ev["particle_id"].to(dev), pt=ev["pt"].to(dev), eta=ev["eta"].to(dev),
                           reconstructable=ev["reconstructable"].to(dev))
        self.cpu_event = {k: v for k, v in ev.items()} | {"x14": x}
        torch.manual_seed(0)
        self.model = G.GraphTCN(14, 28, h_outdim=self.DIM, hidden_dim=40, L_ec=3, L_hc=3, alpha_latent=0.9,
                                n_embedding_coords=self.DIM).to(dev)
        self.preproc = G.MLGraphConstruction(ml=None, embedding_slice=(0, self.DIM), max_radius=1.0,
                                              max_num_neighbors=self.K_NN)
        self.flat = gdist.FlatParameters(self.model)
        self.module = training.TCModule(
            self.model, loss_fct=G.CondensationLossRG(lw_repulsive=1.0, lw_noise=0.1, lw_coward=0.1),
            preproc=self.preproc, flat=self.flat, scheduler=None, bf16=dtype == "bf16",
            optimizer=lambda p: torch.optim.Adam(p, lr=1e-4))
        self.stage = _StageTimer()
        self.n_hits = n
        built = self.preproc(self._fresh())
        self.n_edges = int(built.edge_index.shape[1])
        self.edges_per_step_global = self.n_edges * world
        mask = G.get_good_node_mask_tensors(pt=self.data.pt, particle_id=self.data.particle_id,
                                            reconstructable=self.data.reconstructable, eta=self.data.eta)
        self.n_cp = int(torch.unique(self.data.particle_id[mask]).numel())
        self.info = {"hits": n, "knn_k": self.K_NN, "edges_built": self.n_edges, "condensation_points": self.n_cp}
        self.describe = (f"cfg5: per GPU one pile-up-like event of {n} hits (14 node features, 8-d latent slice: 6000 "
                         f"Gaussian clusters + 10 % noise); step = MLGraphConstruction (kNN k={self.K_NN}, r=1 -> "
                         f"{self.n_edges} edges, labels, 28 edge features) + GraphTCN(14, 28, h_outdim=8, hidden 40, "
                         f"L_ec=3, L_hc=3, alpha_latent=0.9) forward + CondensationLossRG (K={self.n_cp}) + backward + Adam")

    def _fresh(self):
        import copy
        return copy.copy(self.data)

    def step(self):
        m, st = self.module, self.stage
        opt = m.configure_optimizers()
        m.zero_grad()
        ops.clear_graph_index_cache()
        with st("graph_build"):
            data = m.data_preproc(self._fresh())
        with bf16_storage(m.bf16):   # (what TrackingModule.backward_step wraps around the same calls)
            with st("model_forward"):
                out = m(data, _preprocessed=True)
            with st("oc_loss_forward"):
                loss, _ = m.get_losses(out, data, metrics=False)
            with st("backward"), ops.grad_sinks_armed():
                loss.backward()
        with st("allreduce_adam"):
            self.flat.all_reduce_grads()
            opt.step()
        return loss.detach()

    def stages(self) -> dict:
        s = self.stage.summary()
        n, d, k = self.n_hits, self.DIM, self.n_cp
        if "graph_build" in s:
            # the pruned search visits a data-dependent few per cent of the N^2 pairs, so there is no
            # fixed flop count to price it with: the roofline block is the MEASURED vector-pipe
            # utilisation of the search kernel (committed SQ pass); the brute-force-equivalent rate
            # (what an exhaustive N^2 D search would need to run at to finish in the stage's time) is
            # kept as a separately named figure
            t = s["graph_build"]["avg_ms"] * 1e-3
            roof = measured_pipe("cfg5", "knn_pruned_kernel")
            if roof is not None:
                roof["note"] = ("dominant kernel of the stage (pruned exact search, bit-identical to brute force); the "
                                "rest of the stage: sort / boxes / emit / labels / edge features (HBM-bound tails)")
                s["graph_build"]["roofline"] = roof
            s["graph_build"]["bruteforce_equivalent_tflops"] = 2.0 * n * n * d / t / 1e12
        if "oc_loss_forward" in s:
            t = s["oc_loss_forward"]["avg_ms"] * 1e-3
            roof = measured_pipe("cfg5", "oc_hits_spatial_kernel")
            if roof is not None:
                roof["note"] = ("hit pass of the spatial condensation-loss forward; the stage also holds the "
                                "condensation-point selection (sort + scans) and the chunk build")
                s["oc_loss_forward"]["roofline"] = roof
            s["oc_loss_forward"]["dense_equivalent_tflops"] = float(n) * k * (3 * d + 12) / t / 1e12
        return s

    def roofline(self, ks):
        roof, kernels = roofline_of(ks, self.dtype)
        return roof, kernels

    def cpu_baseline(self, iters: int) -> dict:
        """The oracle's TC step on the first 20 000 hits of the event (the reference's kNN is
        O(N^2) on the CPU as well: the full event would take minutes)."""
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import ref_cpu as O

        ns = min(20_000, self.n_hits)
        ev = self.cpu_event
        raw = {"x": ev["x14"][:ns], "particle_id": ev["particle_id"][:ns], "pt": ev["pt"][:ns],
               "eta": ev["eta"][:ns], "reconstructable": ev["reconstructable"][:ns]}
        p = {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()}
        threads = min(32, os.cpu_count() or 1)
        torch.set_num_threads(threads)
        kw = dict(mlgc=dict(embedding_slice=(0, self.DIM), max_radius=1.0, max_num_neighbors=self.K_NN),
                  gtcn=dict(L_ec=3, L_hc=3, alpha_latent=0.9, n_embedding_coords=self.DIM), loss_kind="rg",
                  loss_weights=(1.0, 0.1, 0.1))
        times, edges = [], 0
        for _ in range(max(1, min(iters, 2))):
            t0 = time.perf_counter()
            graph, *_ = O.tc_training_step(raw, p, **kw)
            times.append(time.perf_counter() - t0)
            edges = int(graph["edge_index"].shape[1])
        dt = min(times)
        return {"value": edges / dt, "unit": "edges/s", "cores": threads, "kind": "port",
                "sample": f"oracle TC step (kNN + GraphTCN + CondensationLossRG + backward + Adam) on the first {ns} "
                          f"hits of the event ({edges} edges), best of {len(times)}, {threads} threads",
                "s_per_iter": dt, "hits_per_s": ns / dt}


class StubWorkload(Workload):
    """TEST ONLY (``--stub``): a CPU toy model so that tests can drive the launcher, barrier,
    max-over-ranks timing and the all-reduce through this file's real control flow."""

    name, scaling = "stub", "weak"

    def __init__(self, args, rank: int, world: int, dev):
        torch.manual_seed(0)
        self.model = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.ReLU(), torch.nn.Linear(4, 1))
        self.flat = gdist.FlatParameters(self.model)
        self.opt = torch.optim.Adam([self.flat.flat_param], lr=1e-2)
        g = torch.Generator().manual_seed(100 + rank)   # per-rank data
        self.x = torch.randn(64, 3, generator=g)
        self.edges_per_step_global = 64 * world
        self.describe = "stub: CPU toy model (launcher / collective control-flow test), not a measurement"
        self.info = {"rank_seed": 100 + rank}

    def step(self):
        self.flat.zero_grad()
        loss = self.model(self.x).pow(2).mean()
        loss.backward()
        self.flat.all_reduce_grads()
        self.opt.step()
        return loss.detach()


# ------------------------------------------------------------------------------ timing
def barrier(world: int) -> None:
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_steps(wl: Workload, world: int, dev, steps: int, warmup: int, *, kernel_timer: bool):
    """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides;
    returns (seconds = max over ranks, last loss, kernel summary)."""
    # the per-kernel / per-stage HIP-event brackets are armed during the warm-up as well (their
    # first use costs tens of milliseconds of host time inside the HIP runtime - a one-time cost
    # like the first launch of a kernel) and emptied before the timed region
    timer = ops.KernelTimer() if kernel_timer and not os.environ.get("GNNTRK_BENCH_NO_TIMER") else None
    ops.set_kernel_timer(timer)
    if hasattr(wl, "stage"):
        wl.stage.on = True
    for i in range(warmup):
        wl.step()
        if i == warmup - 2:
            # one synchronisation INSIDE the warm-up: on a fresh box the first step after the process's
            # first device synchronisation costs its host thread 130 ms (measured: the runtime defers
            # one-time work to that point); the last warm-up step now pays it, not the first timed one
            barrier(world)
    # the timed region's events are created NOW (the warm-up has shown how many a step takes): the
    # HIP runtime grows its event storage in steps, and such a step inside the timed region costs
    # its host thread tens of milliseconds - which a 9 ms cfg5 step cannot hide
    used = 0
    if timer is not None:
        used += 2 * len(timer.records)
        timer.records.clear()
    if hasattr(wl, "stage"):
        used += 2 * len(wl.stage.rec)
        wl.stage.rec.clear()
    if used and dev.type == "cuda":
        ops.reserve_timing_events((used // max(warmup, 1) + 8) * steps)
    debug = bool(os.environ.get("GNNTRK_BENCH_DEBUG")) and dev.type == "cuda"
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if debug else None
    barrier(world)
    t0 = time.perf_counter()
    marks = []
    if debug:
        evs[0].record()
    for i in range(steps):
        loss = wl.step()
        marks.append(time.perf_counter())   # (host-side enqueue times: no synchronisation)
        if debug:
            evs[i + 1].record()
    barrier(world)
    dt = time.perf_counter() - t0
    if debug:
        print("host ms per step:", [round((b - a) * 1e3, 1) for a, b in zip([t0] + marks, marks)], file=sys.stderr)
        print("gpu  ms per step:", [round(a.elapsed_time(b), 1) for a, b in zip(evs, evs[1:])], file=sys.stderr)
    ops.set_kernel_timer(None)
    if hasattr(wl, "stage"):
        wl.stage.on = False
    global PER_RANK_MS
    PER_RANK_MS = [dt / steps * 1e3]
    if world > 1:
        # every rank's own wall time between the two barriers (they differ by how long a rank waited in the
        # closing barrier: the spread shows load imbalance); the job's time is the maximum
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[torch.distributed.get_rank()] = dt
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        PER_RANK_MS = [float(v) / steps * 1e3 for v in t.tolist()]
        dt = float(t.max().item())
    return dt, float(loss.item()), (timer.summary() if timer else {})


PER_RANK_MS: list = []


def collective_info(dist_backend: str) -> dict:
    """What the first multi-GPU line should say about itself (rank 0, N > 1): the collective library's version
    and how the GPUs of the node are linked (``rocm-smi --showtopo``: link type and hop count per pair), so
    that a scaling curve can be read against the xGMI topology without a second run.  Best effort: anything
    that cannot be probed is null."""
    import shutil
    import subprocess

    info: dict = {"rccl_version": None, "link_type": None, "hops": None}
    if dist_backend == "nccl":
        try:
            v = torch.cuda.nccl.version()
            info["rccl_version"] = ".".join(map(str, v)) if isinstance(v, tuple) else str(v)
        except Exception:  # noqa: BLE001
            pass
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        out = subprocess.run([smi, "--showtopo"], capture_output=True, text=True, timeout=30).stdout
    except Exception:  # noqa: BLE001
        return info
    info.update(parse_showtopo(out))
    return info


def parse_showtopo(text: str) -> dict:
    """``rocm-smi --showtopo`` -> {"link_type": [[...]], "hops": [[...]]} (matrices over the GPUs in the tool's
    order; null where a block is missing)."""
    def block(title: str):
        rows, on = [], False
        for ln in text.splitlines():
            if title in ln:
                on = True
                continue
            if on:
                if ln.startswith("=") and rows:
                    break
                parts = ln.split()
                if parts and parts[0].startswith("GPU") and len(parts) > 1 and not parts[1].startswith("GPU"):
                    rows.append(parts[1:])
        return rows or None

    hops = block("Hops between two GPUs")
    if hops is not None:
        try:
            hops = [[int(v) for v in r] for r in hops]
        except ValueError:
            pass
    return {"link_type": block("Link Type between two GPUs"), "hops": hops}


def roofline_of(ks: dict, dtype: str):
    """Per-instantiation table and the roofline block of the dominant fused-MLP kernel."""
    kernels = {}
    for k, d in ks.items():
        kernels[k] = {"launches": d["launches"], "avg_ms": d["ms"] / d["launches"],
                      "tflops": d["flops"] / (d["ms"] * 1e-3) / 1e12,
                      "alg_GBps": d["bytes"] / (d["ms"] * 1e-3) / 1e9}
        # HBM bytes actually moved (committed PMC passes) over THIS run's launch time
        tr = measured_traffic(k, d["rows"] / d["launches"], dtype)
        if tr is not None:
            gbps = tr / (d["ms"] / d["launches"] * 1e-3) / 1e9
            kernels[k].update({"traffic": tr, "hbm_GBps": gbps, "hbm_frac": gbps / (PEAK_HBM_TBPS * 1e3)})
    if not ks:
        return None, kernels
    dom = max(ks, key=lambda k: ks[k]["ms"])
    d = ks[dom]
    tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
    gbps = d["bytes"] / (d["ms"] * 1e-3) / 1e9
    common = {"kernel": dom, "launches": d["launches"], "avg_launch_ms": d["ms"] / d["launches"],
              "alg_flops_per_launch": d["flops"] / d["launches"],
              "alg_bytes_per_launch": d["bytes"] / d["launches"],
              "traffic": measured_traffic(dom, d["rows"] / d["launches"], dtype)}
    if common["traffic"] is not None:
        common["traffic_source"] = TRAFFIC_NOTES.get(dom)
    if dtype == "bf16":
        # bf16 MFMA (2.5 PFLOP/s) leaves the fused kernels HBM / issue bound: the
        # roofline that bounds them is HBM bandwidth
        roof = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_TBPS * 1e3, "unit": "GB/s",
                "frac": gbps / (PEAK_HBM_TBPS * 1e3), **common, "mfma_tflops_algorithmic": tf}
    else:
        roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / PEAK_F32_MFMA_TFLOPS, **common, "hbm_frac_algorithmic": gbps / (PEAK_HBM_TBPS * 1e3)}
    return roof, kernels


def hipgraph_cfg2(dev, dtype: str, steps: int, hits: int = 10_000, edges: int = 100_000, seed: int = 1,
                  label: str = "cfg2") -> dict:
    """One event per step (the reference's DataLoader default, utils/loading.py:235: batch_size = 1) eager and as ONE
    captured HIP graph: every gnntrk_* entry point is stream ordered, allocation- and sync-free.  cfg2 (10 k hits:
    launch bound, ~100 kernels of a few microseconds) and one full-size event of cfg3 (150 k hits, 2 M edges)."""
    torch.manual_seed(0)
    model = G.ECForGraphTCN(node_indim=14, edge_indim=4, **EC_MODEL).to(dev)
    # (parameters and gradients in one bucket, as in the headline: one Adam over the bucket and the weight gradients
    #  added by the backward launches themselves - with 45 separate tensors the captured step carried 137 more
    #  launches of a few microseconds each, profiles/r05_one_event_timeline.md)
    flat = gdist.FlatParameters(model)
    mod = training.ECModule(model, loss_fct=G.EdgeWeightBCELoss(), bf16=dtype == "bf16", flat=flat, scheduler=None,
                            optimizer=lambda p: torch.optim.Adam(p, lr=1e-4, weight_decay=1e-4, capturable=True,
                                                                 fused=os.environ.get("GNNTRK_BENCH_FUSED_ADAM", "1") != "0"))
    batch = G.collate([synthetic.make_event(seed, hits, edges, dev)])

    def step():
        ops.clear_graph_index_cache()
        return mod.optimisation_step(batch)

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(5):
        step()
    eager = timed(step, steps)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay()
    graph = timed(g.replay, steps)
    E = batch.num_edges
    return {"workload": f"{label}: 1 event x {hits} hits x {edges} edges per step (graph index + forward + BCE + backward "
                        "+ Adam: one fused kernel over the parameter bucket), eager and the whole step as one hipGraph replay",
            "dtype": dtype, "steps": steps, "eager_ms_per_step": eager, "ms_per_step": graph,
            "value": E / graph * 1e3, "eager_value": E / eager * 1e3, "unit": "edges/s",
            "roofline": "n/a (cache resident, launch bound)" if edges <= 200_000 else
                        "see the headline's roofline block (same kernels, 1/32 of the rows per launch)"}


def cfg4_short(args, rank: int, world: int, dev) -> dict:
    """BASELINE config 4 (strong scaling over the same 256 events) as a short run; collective:
    every rank calls it."""
    wl = ECWorkload(args, rank, world, dev, workload="cfg4", dtype="bf16")
    dt, loss, _ = timed_steps(wl, world, dev, 3, 1, kernel_timer=False)
    return {"workload": wl.describe, "n_gpus": world, "steps": 3, "warmup": 1, "ms_per_step": dt / 3 * 1e3,
            "value": wl.edges_per_step_global * 3 / dt, "unit": "edges/s", "scaling": "strong",
            "final_loss": loss, **wl.info}


def cfg5_short(args, dev, dtype: str = "f32") -> dict:
    """BASELINE config 5 as a short run: the object-condensation step on one 200 000-hit event with
    per-stage HIP-event times (kNN graph build, GraphTCN forward, condensation loss, backward)."""
    import copy
    a5 = copy.copy(args)
    a5.events = None
    wl = TCWorkload(a5, 0, 1, dev, dtype=dtype)
    dt, loss, _ = timed_steps(wl, 1, dev, 10, 3, kernel_timer=False)
    return {"workload": wl.describe, "steps": 10, "warmup": 3, "ms_per_step": dt / 10 * 1e3, "dtype": dtype,
            "value": wl.edges_per_step_global * 10 / dt, "unit": "edges/s", "hits_per_s": wl.n_hits * 10 / dt,
            "final_loss": loss, "stages": wl.stages(), **wl.info}


def dbscan_short(dev) -> dict:
    """postprocessing.DBSCANFastRescan (fastrescanner.py:6-66) on the cfg5 cloud: ONE radius graph at
    max_eps, then four (eps, min_pts) rescans - what DBSCANHyperParamScannerFast asks per trial."""
    from gnn_tracking_amd.postprocessing import DBSCANFastRescan

    n, max_eps = 200_000, 0.5
    x = synthetic.make_pileup_cloud(500, n).to(dev)
    DBSCANFastRescan(x[:4096], max_eps=max_eps).cluster_device(max_eps, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fr = DBSCANFastRescan(x, max_eps=max_eps)
    torch.cuda.synchronize()
    t_graph = time.perf_counter() - t0
    trials = ((max_eps, 1), (0.6 * max_eps, 2), (0.4 * max_eps, 3), (0.3 * max_eps, 4))
    t0 = time.perf_counter()
    n_clusters = [int(fr.cluster_device(eps, mp).max()) + 1 for eps, mp in trials]
    torch.cuda.synchronize()
    t_clu = (time.perf_counter() - t0) / len(trials)
    flops = 2.0 * n * n * 8 * 3   # count + fill pass: sub, mul, add per dimension in fp64
    out = {"workload": f"DBSCANFastRescan on {n} hits in 8-d, max_eps {max_eps}: radius graph + {len(trials)} rescans",
           "radius_graph_ms": t_graph * 1e3, "radius_graph_edges": int(fr._n_edges),
           "rescan_ms_per_trial": t_clu * 1e3, "clusters_per_trial": n_clusters,
           "bruteforce_equivalent_tflops_fp64": flops / t_graph / 1e12}
    roof = measured_pipe("dbscan", "radius_pruned_kernel")
    if roof is not None:
        roof["note"] = ("pruned walk of the radius graph (identical lists to the exhaustive kernels): measured "
                        "vector-pipe utilisation of its count pass; the fill pass runs the same kernel")
        out["roofline"] = roof
    return out


def gc_resin_short(dev, hits: int = 200_000, edges: int = 3_000_000, steps: int = 10, bf16: bool = True) -> dict:
    """GraphConstructionResIN at the reference's default hidden_dim = 40 (120 -> 40 -> 40 -> 40 relational model):
    forward + backward on one 200 k-hit / 3 M-edge graph - in bf16 storage the output-tile / wide-input kernels, in
    fp32 the wide fused kernels of csrc/mlp_wide.hip."""
    ev = synthetic.make_event(7, hits, edges, dev)
    data = G.Data(x=ev.x, edge_index=ev.edge_index, edge_attr=ev.edge_attr)
    torch.manual_seed(0)
    model = G.GraphConstructionResIN(node_indim=14, edge_indim=4, n_layers=1).to(dev)

    def step():
        model.zero_grad()
        with G.bf16_storage(bf16):
            model(data)["H"].float().square().mean().backward()

    ops._WIDE_WARNED.clear()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"workload": f"GraphConstructionResIN(hidden_dim=40), {hits} hits, {edges} edges, forward + backward, "
                        f"{'bf16 storage' if bf16 else 'fp32'}",
            "steps": steps, "ms_per_step": ms, "value": edges / ms * 1e3, "unit": "edges/s",
            "library_path_taken": sorted(map(str, ops._WIDE_WARNED))}


def extras(args, rank: int, world: int, dev) -> dict:
    """Short driver-timed runs of the configurations the headline does not cover: cfg4 at
    every rank count; at N = 1 also cfg3 in fp32 and cfg2 as a HIP graph."""
    out = {}
    try:
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
        out["cfg4_strong_bf16"] = cfg4_short(args, rank, world, dev)
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
        if world > 1:
            return out
        wl = ECWorkload(args, 0, 1, dev, workload="cfg3", dtype="f32")
        dt, loss, ks = timed_steps(wl, 1, dev, 5, 2, kernel_timer=True)
        roof, _ = roofline_of(ks, "f32")
        out["cfg3_f32"] = {"workload": "cfg3 in the reference's precision (fp32 storage, fp32 MFMA)", "steps": 5,
                           "warmup": 2, "ms_per_step": dt / 5 * 1e3, "value": wl.edges_per_step_global * 5 / dt,
                           "unit": "edges/s", "final_loss": loss, "roofline": roof}
        del wl
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
        wl = ECWorkload(args, 0, 1, dev, workload="cfg3", dtype="bf16", index="resident")
        dt, loss, _ = timed_steps(wl, 1, dev, 5, 2, kernel_timer=False)
        out["cfg3_bf16_resident_index"] = {
            "workload": "cfg3 with the graph index (and the CSR-ordered labels) of the batch kept resident in HBM "
                        "instead of rebuilt in every step - epochs >= 2 over a dataset that stays on the device; NOT "
                        "the headline (which pays the index in every step)",
            "steps": 5, "warmup": 2, "ms_per_step": dt / 5 * 1e3, "value": wl.edges_per_step_global * 5 / dt,
            "unit": "edges/s", "final_loss": loss}
        del wl
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
        if getattr(args, "node_ids", "random") != "phi":
            import copy
            a_phi = copy.copy(args)
            a_phi.node_ids = "phi"
            wl = ECWorkload(a_phi, 0, 1, dev, workload="cfg3", dtype="bf16")
            dt, loss, ks = timed_steps(wl, 1, dev, 5, 2, kernel_timer=True)
            roof, _ = roofline_of(ks, "bf16")
            out["cfg3_bf16_phi_ids"] = {
                "workload": "cfg3 with the hits of every event numbered by phi and edges joining ids at most 64 apart "
                            "(synthetic.make_event(phi_sorted_ids=True)): the best case for the locality of the node-row "
                            "gathers and the source-sorted gradient stores; the headline's generator shuffles the ids "
                            "(worst case); same model, same timed region - NOT the headline",
                "steps": 5, "warmup": 2, "ms_per_step": dt / 5 * 1e3, "value": wl.edges_per_step_global * 5 / dt,
                "unit": "edges/s", "final_loss": loss, "roofline": roof}
            del wl
            ops.clear_graph_index_cache()
            torch.cuda.empty_cache()
        from gnn_tracking_amd import locality
        if locality.mode() != "off":
            with G.node_order("off"):
                wl = ECWorkload(args, 0, 1, dev, workload="cfg3", dtype="bf16")
                dt, loss, ks = timed_steps(wl, 1, dev, 5, 2, kernel_timer=True)
            roof, _ = roofline_of(ks, "bf16")
            out["cfg3_bf16_node_order_off"] = {
                "workload": "cfg3 exactly as the headline but with the node renumbering switched off (GNNTRK_NODE_ORDER=off: "
                            "the graph index and every gather in the generator's shuffled numbering, as in rounds 1-4)",
                "steps": 5, "warmup": 2, "ms_per_step": dt / 5 * 1e3, "value": wl.edges_per_step_global * 5 / dt,
                "unit": "edges/s", "final_loss": loss, "roofline": roof}
            del wl
            ops.clear_graph_index_cache()
            torch.cuda.empty_cache()
        wl = ECWorkload(args, 0, 1, dev, workload="cfg3", dtype="bf16", cached_index=True)
        dt, loss, ks = timed_steps(wl, 1, dev, 5, 2, kernel_timer=False)
        out["cfg3_bf16_cached_index"] = {
            "workload": "cfg3 with the graph index of every EVENT built once and kept on the device (labels / edge features "
                        "carried, node order per event); every step COLLATES the cached indices into the batch's arrays "
                        "(ops.place_graph_indices: one streaming pass per event, identical arrays) instead of sorting the "
                        "batch - epochs >= 2 over a static dataset (utils/loading.py:97-100); the per-event builds are "
                        "AMORTISED, not the headline (which sorts every batch inside the step)",
            "steps": 5, "warmup": 2, "ms_per_step": dt / 5 * 1e3, "value": wl.edges_per_step_global * 5 / dt,
            "unit": "edges/s", "final_loss": loss}
        del wl
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
        wl = ECWorkload(args, 0, 1, dev, workload="cfg3", dtype="bf16", resident_dataset=True)
        dt, loss, ks = timed_steps(wl, 1, dev, 5, 2, kernel_timer=False)
        out["cfg3_bf16_resident_dataset"] = {
            "workload": "cfg3 through io.ResidentDataset: the 32 events stay on the device with their per-event index "
                        "(built once: AMORTISED, not the headline); EVERY step draws them in a new order, collates them "
                        "on the device (as the reference's DataLoader collates on the host) and places the indices - "
                        "collation + placement are inside the timed step, nothing is sorted",
            "steps": 5, "warmup": 2, "ms_per_step": dt / 5 * 1e3, "value": wl.edges_per_step_global * 5 / dt,
            "unit": "edges/s", "final_loss": loss}
        del wl
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
        wl = ECWorkload(args, 0, 1, dev, workload="cfg3", dtype="bf16", loader_renumbered=True)
        dt, loss, ks = timed_steps(wl, 1, dev, 5, 2, kernel_timer=True)
        roof, _ = roofline_of(ks, "bf16")
        out["cfg3_bf16_loader_renumbered"] = {
            "workload": "cfg3 with every event renumbered ONCE by the loader-side transform (io.renumber_nodes, what "
                        "GraphDataset(renumber=True) does when a graph is read) OUTSIDE the timed steps - the cost a static "
                        "dataset pays once per event, AMORTISED here, not the headline (whose batches arrive in the "
                        "generator's shuffled order and are renumbered inside every step); same generator, model and timed region",
            "steps": 5, "warmup": 2, "ms_per_step": dt / 5 * 1e3, "value": wl.edges_per_step_global * 5 / dt,
            "unit": "edges/s", "final_loss": loss, "roofline": roof}
        del wl
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
        wl = ECWorkload(args, 0, 1, dev, workload="cfg3", dtype="bf16", hidden_dim=64)
        dt, loss, ks = timed_steps(wl, 1, dev, 5, 2, kernel_timer=True)
        out["cfg3_hidden64_bf16"] = {
            "workload": "cfg3 with ECForGraphTCN(hidden_dim=64): four hidden tiles, biases as accumulator initial "
                        "values (mlp16_*_bi_kernel); before round 3 this width took library GEMMs",
            "steps": 5, "warmup": 2, "ms_per_step": dt / 5 * 1e3, "value": wl.edges_per_step_global * 5 / dt,
            "unit": "edges/s", "final_loss": loss, "kernels": sorted(ks)}
        del wl
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
        out["gc_resin_default_bf16"] = gc_resin_short(dev)
        out["gc_resin_default_f32"] = gc_resin_short(dev, bf16=False)
        out["cfg2_hipgraph_bf16"] = hipgraph_cfg2(dev, "bf16", 100)
        out["cfg2_hipgraph_f32"] = hipgraph_cfg2(dev, "f32", 100)
        # the reference's default operating point: ONE full-size event per step (utils/loading.py:235)
        out["one_event_150k_2M_bf16"] = hipgraph_cfg2(dev, "bf16", 50, 150_000, 2_000_000, 100, "one event of cfg3")
        out["one_event_150k_2M_f32"] = hipgraph_cfg2(dev, "f32", 30, 150_000, 2_000_000, 100, "one event of cfg3")
        torch.cuda.empty_cache()
        out["cfg5_oc_step_f32"] = cfg5_short(args, dev)
        out["cfg5_oc_step_bf16"] = cfg5_short(args, dev, "bf16")
        out["dbscan_rescan_200k"] = dbscan_short(dev)
    except Exception as e:  # the headline line must survive a failing extra
        if world > 1:
            raise   # (a rank that drops out of a collective would hang the others)
        out["error"] = f"{type(e).__name__}: {e}"
    return out



# ------------------------------------------------------------------------------- the line
LINE_LIMIT = 6000          # bytes of the final stdout line (the driver parses that line; round 5's 22 KB line was lost)
EXTRA_FILE = "bench_extra.json"
_SIDE_KEYS = ("s_all_iters", "thread_probe_s", "oracle_seconds", "alg_flops_per_launch", "what", "oracle", "event")


def _short(v, n: int = 6):
    """Floats to n significant digits, recursively (a line of 17-digit floats is a third longer for nothing)."""
    if isinstance(v, float):
        return float(f"{v:.{n}g}")
    if isinstance(v, dict):
        return {k: _short(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_short(x, n) for x in v]
    return v


def _drop(d, keys):
    if isinstance(d, dict):
        return {k: _drop(v, keys) for k, v in d.items() if k not in keys}
    return d


def extra_digest(extra: dict | None) -> dict | None:
    """name -> the few numbers of a side run that belong next to the headline (everything else: EXTRA_FILE)."""
    if not extra:
        return None
    out = {}
    for k, v in extra.items():
        if not isinstance(v, dict):
            out[k] = v if not isinstance(v, str) else v[:200]
            continue
        d = {f: v[f] for f in ("ms_per_step", "eager_ms_per_step", "dtype", "n_gpus", "radius_graph_ms", "rescan_ms_per_trial",
                               "launches_per_step") if f in v}
        if isinstance(v.get("roofline"), dict) and "frac" in v["roofline"]:
            d["roofline_frac"] = v["roofline"]["frac"]
        if isinstance(v.get("stages"), dict):
            d["stage_ms"] = {s: t.get("avg_ms") for s, t in v["stages"].items() if isinstance(t, dict)}
        out[k] = d
    return out


def kernel_digest(kernels: dict | None) -> dict | None:
    """kernel -> [avg ms per launch, algorithmic GB/s, HBM fraction from the committed traffic pass or null]."""
    if not kernels:
        return None
    return {k: [d["avg_ms"], d["alg_GBps"], d.get("hbm_frac")] for k, d in kernels.items()}


def emit(line: dict, kernels: dict | None, extra: dict | None) -> None:
    """Everything measured goes to EXTRA_FILE (next to bench.py and, where the directory exists, under gpurun_out/)
    and to stderr as one line; stdout gets exactly ONE line of at most LINE_LIMIT bytes: the contract's keys,
    ``roofline`` (with the access floors), ``cpu_baseline``, ``parity_check``, ``library``, ``stages`` and a digest
    of the per-kernel table and of the side runs."""
    full = dict(line, kernels=kernels, extra=extra)
    blob = json.dumps(full)
    for path in (os.path.join(ROOT, EXTRA_FILE), os.path.join(ROOT, "gpurun_out", EXTRA_FILE)):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(blob + "\n")
        except OSError:
            pass
    print("bench_extra " + blob, file=sys.stderr, flush=True)
    out = _short(_drop(line, _SIDE_KEYS))
    out["extra_file"] = EXTRA_FILE
    out["kernels"] = _short(kernel_digest(kernels), 4)
    out["extra"] = _short(extra_digest(extra), 5)
    for victim in ("kernels", "extra", "stages", "parity_check"):   # (never needed so far; the line must stay parseable)
        if len(json.dumps(out)) <= LINE_LIMIT:
            break
        out[victim] = f"see {EXTRA_FILE}"
    text = json.dumps(out)
    assert len(text) <= LINE_LIMIT or line.get("data") == "stub", f"bench line is {len(text)} bytes"
    print(text, flush=True)


# ----------------------------------------------------------------------------------- main
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))

    backend = "gloo" if args.stub else args.backend
    if backend == "gloo" and not args.stub and torch.cuda.is_available():
        # (ranks may share a device under gloo: map them round-robin before the group is built)
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    rank, local, world = gdist.init_process_group_from_env(
        backend=backend, timeout_s=float(os.environ.get("GNNTRK_DIST_TIMEOUT_S", "180")))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has WORLD_SIZE={world}")
    if world > 1:
        assert torch.distributed.is_initialized() and torch.distributed.get_world_size() == args.gpus
    # the collective library really in use: "nccl" (= RCCL on ROCm) / "gloo"; "none" for one process
    dist_backend = torch.distributed.get_backend() if world > 1 else "none"
    if args.stub:
        dev = torch.device("cpu")
        wl: Workload = StubWorkload(args, rank, world, dev)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (the package has no CPU path)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if args.workload == "cfg5":
            wl = TCWorkload(args, rank, world, dev, dtype=args.dtype)
        else:
            wl = ECWorkload(args, rank, world, dev, workload=args.workload, dtype=args.dtype, index=args.index)

    dt, final_loss, ks = timed_steps(wl, world, dev, args.steps, args.warmup, kernel_timer=not args.stub)
    # sum of |parameters| after the run: equal (to rounding) at every rank count for cfg4, whose
    # step is the same arithmetic however the shards are spread over ranks
    param_checksum = (float(sum(p.detach().double().abs().sum() for p in wl.model.parameters()))
                      if hasattr(wl, "model") else None)
    want_extra = not args.no_extra and not args.stub and args.workload == "cfg3" and args.dtype == "bf16"
    model_for_cpu = (getattr(wl, "first_event_cpu", None), getattr(wl, "model", None))
    describe, info, scaling, wdtype = wl.describe, wl.info, wl.scaling, getattr(wl, "dtype", args.dtype)
    edges_per_step = wl.edges_per_step_global
    roof_fn = getattr(wl, "roofline", None)
    cpu_fn = getattr(wl, "cpu_baseline", None)
    stages = wl.stages() if hasattr(wl, "stages") else None
    floor = None
    if (rank == 0 and world == 1 and not args.stub and isinstance(wl, ECWorkload) and args.workload == "cfg3"
            and args.dtype == "bf16" and not args.no_extra):
        try:
            floor = access_floor(wl, dev)
        except Exception as e:   # (the headline survives a failing side measurement)
            floor = {"error": f"{type(e).__name__}: {e}"}
        ops.clear_graph_index_cache()
        torch.cuda.empty_cache()
    extra = None
    if want_extra:
        del wl
        extra = extras(args, rank, world, dev)

    if rank == 0:
        if roof_fn is not None:
            roof, kernels = roof_fn(ks)
        else:
            roof, kernels = roofline_of(ks, args.dtype)
        if roof is not None and floor is not None:
            roof["access_floor"] = floor
        cpu = parity = None
        if world == 1 and not args.no_cpu_baseline and not args.stub:
            if cpu_fn is not None:
                cpu = cpu_fn(args.cpu_iters)
            elif model_for_cpu[0] is not None:
                cpu = cpu_baseline(model_for_cpu[0], model_for_cpu[1], args.cpu_iters)
                try:
                    parity = parity_check(model_for_cpu[0], model_for_cpu[1], wdtype, dev)
                except Exception as e:
                    parity = {"ok": False, "error": f"{type(e).__name__}: {e}"}
        total = edges_per_step * args.steps
        line = {
            "metric": "edges_per_sec_fwd_bwd" if not args.stub else "stub_not_a_measurement",
            "value": total / dt,
            "unit": "edges/s",
            "n_gpus": world,
            "ranks": world,
            "backend": dist_backend,
            **({"rccl_ranks": torch.distributed.get_world_size()} if dist_backend == "nccl" else {}),
            "self_launched": bool(os.environ.get("GNNTRK_BENCH_SELF_LAUNCHED")),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            **({"ms_per_step_per_rank": PER_RANK_MS, "collective": collective_info(dist_backend)} if world > 1 else {}),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": wdtype,
            "data": "synthetic" if not args.stub else "stub",
            "config": {
                "workload": describe,
                "graph_index": args.index,
                "global_edges_per_step": edges_per_step,
                "parallelism": (f"dp{world} (events sharded, flat-gradient all-reduce over "
                                + {"nccl": "RCCL", "gloo": "gloo (host; test configuration)"}.get(dist_backend, dist_backend)
                                + ")" if world > 1 else "dp1 (single process, no collective)"),
                **info,
            },
            "timed_region": "graph index + forward + loss + backward + grad all-reduce + optimizer step",
            "edge_layers_per_sec": total * EC_MODEL["L_ec"] / dt,
            "final_loss": final_loss,
            "param_checksum": param_checksum,
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity_check": parity,
            "library": library_id() if not args.stub else None,
        }
        if stages is not None:
            line["stages"] = stages
        emit(line, kernels, extra)
    barrier(world)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
