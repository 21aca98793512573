"""Tensor-level operators of the hot path: thin, autograd-aware wrappers over the C ABI.

PyTorch is plumbing here (device memory, streams, autograd graph); all arithmetic
happens in the hand-written HIP kernels of ``libgnntrk.so``.

Operators
---------
``GraphIndex`` / ``graph_index(edge_index, n_nodes)``
    target-sorted (CSR) + source-sorted view of a COO ``edge_index``; replaces the
    gather/scatter bookkeeping of PyG ``MessagePassing.propagate``
    (reference ``models/interaction_network.py:67``).
``fused_mlp(segs, weights, biases, ...)``
    gather + concat + (2|3)-layer MLP + epilogue in one kernel, custom backward with
    recompute (``models/mlp.py:59-62`` and its call sites).
``segment_sum(rows, gi, by)``
    per-node sum of edge rows (PyG ``aggr="add"``), deterministic.
``permute_rows(x, idx, scatter)``
    COO order <-> CSR order of edge tensors.
``bce_loss(w, y, ...)``
    ``metrics/losses/ec.py:95-121``.
"""

from __future__ import annotations

import os

import ctypes as C
import contextlib
import dataclasses
import weakref
from typing import Optional, Sequence

import torch
from torch import Tensor

from . import _capi

__all__ = [
    "GraphIndex", "graph_index", "Seg", "fused_mlp", "segment_sum", "permute_rows",
    "axpby", "bce_loss", "knn_graph", "edge_labels", "edge_features",
]


def _stream(t: Tensor):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None  # only reachable in the emulator tests (tests/emul)


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _as_rows(t: Tensor) -> Tensor:
    """[M] -> [M,1]; make the feature axis dense (row stride may be anything)."""
    if t.dim() == 1:
        t = t.unsqueeze(1)
    if t.dim() != 2:
        raise ValueError(f"expected a 1-D or 2-D tensor, got shape {tuple(t.shape)}")
    if t.dtype != torch.float32:
        raise TypeError(f"gnn_tracking_amd kernels are fp32; got {t.dtype}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _row_stride(t: Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def _ws(nbytes: int, like: Tensor) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


# ------------------------------------------------------------ kernel timing hook
class KernelTimer:
    """Optional per-launch timing of the fused-MLP kernels with HIP events on the
    stream the kernels are launched on (bench.py's roofline leg).  Each record carries
    the kernel instantiation key and the launch's ALGORITHMIC flops and bytes."""

    def __init__(self):
        self.records = []  # (key, ev0, ev1, flops, bytes, rows)

    def summary(self) -> dict:
        out: dict = {}
        for key, e0, e1, fl, by, rows in self.records:
            d = out.setdefault(key, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, rows=0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
            d["rows"] += rows
        return out


_TIMER: Optional[KernelTimer] = None

# Timing events come from a pool that can be filled ahead of a timed region: the HIP runtime grows
# its event storage in steps, and a step costs the host tens of milliseconds (bench.py)
_EVENT_POOL: list = []


def reserve_timing_events(n: int) -> None:
    """Create ``n`` timing events now and record each TWICE: the first record is what allocates an
    event, and recording an event that has been recorded before is a path of its own in the HIP
    runtime - on a freshly started box its first execution cost the first timed step of bench.py
    130 ms of host time (the code is paged in on first use)."""
    while len(_EVENT_POOL) < n:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        e.record()
        _EVENT_POOL.append(e)


def timing_event():
    return _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)


def set_kernel_timer(t: Optional[KernelTimer]) -> None:
    global _TIMER
    _TIMER = t


def kernel_key(lib, args, backward: bool, bf16: bool = False) -> str:
    """Name of the kernel instantiation the C launcher dispatches to (mlp.hip), exactly as
    rocprofv3 prints it.  Only evaluated while a KernelTimer is installed."""
    if _TIMER is None:
        return ""
    buf = C.create_string_buffer(160)
    _capi.check(lib.gnntrk_mlp_kernel_name(C.byref(args.mlp), args.n_seg, args.seg,
                                           (1 if backward else 0) + (2 if bf16 else 0), buf,
                                           len(buf)), lib)
    return buf.value.decode()


def _mlp_flops_per_row(m) -> int:
    mid = m.hidden * m.hidden if m.n_layers == 3 else 0
    return 2 * (m.in_dim * m.hidden + mid + m.hidden * m.out_dim)


class _timed:
    def __init__(self, t: Tensor, key: str, flops: float, nbytes: float, rows: int):
        self.on = _TIMER is not None and t.is_cuda
        self.args = (key, flops, nbytes, rows)

    def __enter__(self):
        if self.on:
            self.e0 = timing_event()
            self.e1 = timing_event()
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            key, fl, by, rows = self.args
            _TIMER.records.append((key, self.e0, self.e1, fl, by, rows))


# ------------------------------------------------------------------ graph index
@dataclasses.dataclass
class GraphIndex:
    """Device-resident index of one COO edge list (see include/gnntrk.h)."""

    n_nodes: int
    n_edges: int
    perm: Tensor      # [E] int32: CSR position -> original edge id
    tgt: Tensor       # [E] int32 targets in CSR order (sorted)
    src: Tensor       # [E] int32 sources in CSR order
    rowptr_t: Tensor  # [N+1] int32
    rowptr_s: Tensor  # [N+1] int32
    spos: Tensor      # [E] int32: source-sorted order -> CSR position
    spos_inv: Tensor  # [E] int32: CSR position -> position in the source-sorted order
    ready: object = None  # event of a build on a side stream (prefetch_graph_index), joined on first use
    # node renumbering (graph_index(order_by=...)): tgt / src / rowptr_* are in the NEW numbering
    node_perm: Optional[Tensor] = None   # [N] int32: new id -> caller's id (gather node inputs through it)
    node_rank: Optional[Tensor] = None   # [N] int32: caller's id -> new id (gather node results through it)
    order_sig: object = None
    placed: bool = False   # collated from cached per-event indices (place_graph_indices): the loader chose the node order

    def node_values(self, t: Tensor) -> Tensor:
        """Per-node values of the caller (``pt``, ...) in the numbering of ``tgt`` / ``src``."""
        return t if self.node_perm is None else t.index_select(0, self.node_perm)


_GI_CACHE: dict[int, tuple] = {}


def clear_graph_index_cache() -> None:
    _GI_CACHE.clear()


_VALIDATE = bool(os.environ.get("GNNTRK_VALIDATE"))
#: bit 0: library radix-sort form of the graph index (tests, measurements; identical arrays)
_GI_FLAGS = int(os.environ.get("GNNTRK_GI_FLAGS", "0"))
#: per-edge inputs ride along in the graph-index build (False: gathered through perm afterwards;
#: tests, measurements - identical results)
CARRY = os.environ.get("GNNTRK_GI_CARRY", "1") != "0"


def node_order(x: Tensor, col: int, batch: Optional[Tensor] = None, n_events: int = 0):
    """``(perm, rank)`` int32 ``[N]``: the nodes of every event (``batch``: int64 ``[N]``, non-decreasing; None =
    one event; ``n_events``: number of events if known - fewer radix passes) sorted by ``x[:, col]`` (fp32), ties
    in the old order (gnntrk_node_order)."""
    _capi.require_device(x)
    if x.dtype != torch.float32 or x.dim() != 2:
        raise TypeError("node_order: x must be fp32 [N, F]")
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        x = x.contiguous()   # (a transposed / expanded view: the key column is read through the row stride)
    lib = _capi.load()
    n = int(x.shape[0])
    key = x[:, col]
    if batch is not None:
        if batch.dtype != torch.int64 or batch.numel() != n or batch.device != x.device:
            raise ValueError("node_order: batch must be int64 [N] on the device of x")
        batch = batch.contiguous()
    perm = torch.empty(n, dtype=torch.int32, device=x.device)
    rank = torch.empty(n, dtype=torch.int32, device=x.device)
    ws = _ws(lib.gnntrk_node_order_workspace_bytes(n), x)
    _capi.check(lib.gnntrk_node_order(_p(key), int(x.stride(0)), _p(batch), int(n_events), n, _p(perm), _p(rank), _p(ws), ws.numel(),
                                      _stream(x)), lib)
    return perm, rank


def _order_sig(order_by):
    if order_by is None:
        return None
    x, col, batch = order_by[:3]
    return (id(x), x._version, int(col), None if batch is None else (id(batch), batch._version))


def graph_index(edge_index: Tensor, n_nodes: int, *, cache: bool = True,
                validate: Optional[bool] = None, flags: Optional[int] = None,
                carry_label: Optional[Tensor] = None, carry_rows: Optional[Tensor] = None,
                order_by: Optional[tuple] = None) -> GraphIndex:
    """Build (or fetch) the index of ``edge_index`` ([2,E] int64, unsorted COO).

    Cached per tensor OBJECT (weakref + version counter), so the L layers of a
    ResIN stack and the forward/backward of one step share one build.

    ``validate`` (default: the environment variable ``GNNTRK_VALIDATE``): read back the
    build's count of node ids outside ``[0, n_nodes)`` (sources and targets) and raise
    ``IndexError`` as torch's ``index_select`` would.  It costs a host synchronisation, so it
    is off by default; ids out of range then give undefined results.

    ``carry_label`` (1-byte ``[E]``: the dataset's bool ``y``) / ``carry_rows`` (fp32 ``[E, 4]``:
    ``edge_attr``): per-edge inputs that ride along into CSR order INSIDE the build
    (``gnntrk_graph_index_carry``) instead of being gathered through ``perm`` afterwards; the
    results are left on the index (``carried_label`` / ``carried_rows`` below return them).

    ``order_by = (x, col, batch)``: build the index in a renumbering of the nodes - every event's nodes sorted
    by ``x[:, col]`` (``locality.py`` says why) - with ``node_perm`` / ``node_rank`` left on the index.  Only the
    caller that asked for it sees such an index (the cache keeps the two forms apart).
    """
    if edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError(f"edge_index must be [2,E], got {tuple(edge_index.shape)}")
    if edge_index.dtype != torch.int64:
        raise TypeError("edge_index must be int64 (PyG convention)")
    _capi.require_device(edge_index)
    sig = _order_sig(order_by)
    key = (id(edge_index), sig is not None)
    if cache:
        hit = _GI_CACHE.get(key)
        if hit is not None:
            ref, ver, nn, gi = hit
            if (ref() is edge_index and ver == edge_index._version and nn == n_nodes and gi.order_sig == sig
                    and (sig is None or gi._order_ref() is order_by[0])):   # (an id can be reused: same OBJECT)
                return _join(gi)
    lib = _capi.load()
    ei = edge_index.contiguous()
    E = int(ei.shape[1])
    dev = ei.device
    mk = lambda n: torch.empty(n, dtype=torch.int32, device=dev)  # noqa: E731
    gi = GraphIndex(n_nodes, E, mk(E), mk(E), mk(E), mk(n_nodes + 1), mk(n_nodes + 1), mk(E), mk(E))
    d = _capi.GraphIndex(n_nodes, E, _p(gi.perm), _p(gi.tgt), _p(gi.src), _p(gi.rowptr_t),
                         _p(gi.rowptr_s), _p(gi.spos), _p(gi.spos_inv))
    cy = _capi.GraphIndexCarry()
    if order_by is not None:
        if int(order_by[0].shape[0]) != n_nodes:
            raise ValueError("graph_index(order_by): x must hold one row per node")
        gi.node_perm, gi.node_rank = node_order(*order_by)
        gi.order_sig, gi._order_ref = sig, weakref.ref(order_by[0])
        cy.node_rank = _p(gi.node_rank)
    lab = rows = None
    if CARRY and carry_label is not None and E > 0 and _carry_label_ok(carry_label, E, dev):
        lab = carry_label.detach().view(torch.uint8).contiguous().view(-1)
        lab_csr = torch.empty(E, dtype=torch.uint8, device=dev)
        cy.edge_label, cy.label_csr = _p(lab), _p(lab_csr)
    if CARRY and carry_rows is not None and E > 0 and _carry_rows_ok(carry_rows, E, dev):
        from . import ops_bf16
        rows = carry_rows.detach()
        rows_csr = ops_bf16.empty_rows(E, 4, dev)
        cy.edge_rows, cy.rows_csr_bf16 = _p(rows), _p(rows_csr)
        cy.rows_stride, cy.out_stride = int(rows.stride(0)), int(rows_csr.stride(0))
    ws = _ws(lib.gnntrk_graph_index_workspace_bytes_carry(n_nodes, E, int(rows is not None) | (2 if gi.node_rank is not None else 0)), ei)
    _capi.check(lib.gnntrk_graph_index_build_carry(_p(ei), C.byref(d), C.byref(cy), _p(ws), ws.numel(),
                                                   _GI_FLAGS if flags is None else int(flags), _stream(ei)), lib)
    if lab is not None:
        gi._label_csr = (id(carry_label), carry_label._version, weakref.ref(carry_label), lab_csr)
    if rows is not None:
        gi._rows_csr = (id(carry_rows), carry_rows._version, weakref.ref(carry_rows), rows_csr)
    if _VALIDATE if validate is None else validate:
        bad = int(ws[:4].view(torch.int32).item())  # first workspace word: ids out of range
        if bad:
            raise IndexError(f"edge_index: {bad} node ids out of range [0, {n_nodes})")
    gi._built_from = (weakref.ref(edge_index), edge_index._version)
    if cache:
        _cache_put(edge_index, n_nodes, gi)
    return gi


def place_graph_indices(parts: Sequence[GraphIndex], batch, order_col: Optional[int] = None) -> GraphIndex:
    """The graph index of a collated ``batch`` (``data.collate`` of the events the ``parts`` were built from, in that
    order) from CACHED per-event indices: every part is copied into the batch arrays at its node / edge offset
    (gnntrk_graph_index_place) - identical to building the index of ``batch.edge_index``, carried labels / edge
    features and node order included, for one streaming pass instead of two sorts.  For datasets that stay on the
    device across epochs (the reference's are static: utils/loading.py:97-100): build ``graph_index(ev.edge_index,
    ev.num_nodes, cache=False, carry_label=ev.y, carry_rows=ev.edge_attr, order_by=...)`` once per event, collate
    as usual, call this per batch.  The result is registered in the cache under ``batch.edge_index`` (and the
    batch's ``y`` / ``edge_attr`` / ``x``), so ``ECForGraphTCN`` and the losses find it.  Parts without a node order
    among ordered ones (a single-hit event) take the identity order; ``order_col``: the column the loader orders its
    events by - a batch of nothing but unordered parts is then still an ordered batch (identity everywhere)."""
    parts = list(parts)
    if not parts:
        raise ValueError("place_graph_indices: no parts")
    ei = batch.edge_index
    _capi.require_device(ei)
    lib = _capi.load()
    dev = ei.device
    N, E = sum(p.n_nodes for p in parts), sum(p.n_edges for p in parts)
    if int(batch.x.shape[0]) != N or int(ei.shape[1]) != E:
        raise ValueError(f"place_graph_indices: the parts hold {N} nodes / {E} edges, the batch {batch.x.shape[0]} / {ei.shape[1]}")
    mk = lambda n: torch.empty(n, dtype=torch.int32, device=dev)  # noqa: E731
    gi = GraphIndex(N, E, mk(E), mk(E), mk(E), mk(N + 1), mk(N + 1), mk(E), mk(E))
    ordered = [p.node_perm is not None for p in parts]
    lab = [getattr(p, "_label_csr", None) for p in parts]
    rows = [getattr(p, "_rows_csr", None) for p in parts]
    full = [i for i, p in enumerate(parts) if p.n_edges > 0]   # (an event without edges carries nothing)
    has_lab = bool(full) and lab[full[0]] is not None
    has_rows = bool(full) and rows[full[0]] is not None
    if any((lab[i] is not None) != has_lab or (rows[i] is not None) != has_rows for i in full):
        raise ValueError("place_graph_indices: the parts must agree in their carried inputs")
    ptr = getattr(batch, "ptr", None)
    if isinstance(ptr, Tensor) and ptr.numel() == len(parts) + 1 and not ptr.is_cuda:
        sizes = (ptr[1:] - ptr[:-1]).tolist()
        if sizes != [p.n_nodes for p in parts]:
            raise ValueError(f"place_graph_indices: the events of the batch hold {sizes} nodes, the parts {[p.n_nodes for p in parts]}")
    # an event the loader left unordered (fewer than two hits, no key column) among ordered ones is its own order
    # (``order_col``: the loader orders its events by this column - a batch of nothing but such events is still ordered)
    want_order = any(ordered) or order_col is not None
    ident = {}
    if want_order and not all(ordered):
        for i, p in enumerate(parts):
            if not ordered[i]:
                ident[i] = torch.arange(p.n_nodes, dtype=torch.int32, device=dev)
    if want_order:
        gi.node_perm, gi.node_rank = mk(N), mk(N)
    lab_b = torch.empty(E, dtype=torch.uint8, device=dev) if has_lab else None
    rows_b = None
    if has_rows:
        from . import ops_bf16
        rows_b = ops_bf16.empty_rows(E, 4, dev)
    d = _capi.GraphIndex(N, E, _p(gi.perm), _p(gi.tgt), _p(gi.src), _p(gi.rowptr_t), _p(gi.rowptr_s), _p(gi.spos),
                         _p(gi.spos_inv))
    no = eo = 0
    st = _stream(ei)
    for i, (p, l, r) in enumerate(zip(parts, lab, rows)):
        dp = _capi.GraphIndex(p.n_nodes, p.n_edges, _p(p.perm), _p(p.tgt), _p(p.src), _p(p.rowptr_t), _p(p.rowptr_s),
                              _p(p.spos), _p(p.spos_inv))
        pperm, prank = (ident[i], ident[i]) if i in ident else (p.node_perm, p.node_rank)
        _capi.check(lib.gnntrk_graph_index_place(
            C.byref(dp), no, eo, C.byref(d), _p(None if l is None else l[3]), _p(None if l is None else lab_b),
            _p(None if r is None else r[3]), _p(None if r is None else rows_b), _p(pperm), _p(gi.node_perm), _p(prank),
            _p(gi.node_rank), st), lib)
        no += p.n_nodes
        eo += p.n_edges
    y, ea = getattr(batch, "y", None), getattr(batch, "edge_attr", None)
    if lab_b is not None and isinstance(y, Tensor):
        gi._label_csr = (id(y), y._version, weakref.ref(y), lab_b)
    if rows_b is not None and isinstance(ea, Tensor):
        gi._rows_csr = (id(ea), ea._version, weakref.ref(ea), rows_b)
    if want_order:
        col = parts[ordered.index(True)].order_sig[2] if any(ordered) else int(order_col)
        bt = getattr(batch, "batch", None)
        gi.order_sig, gi._order_ref = _order_sig((batch.x, col, bt if isinstance(bt, Tensor) else None)), weakref.ref(batch.x)
    gi._built_from = (weakref.ref(ei), ei._version)
    gi.placed = True
    _cache_put(ei, N, gi)
    return gi


def placed_graph_index(edge_index: Tensor, n_nodes: int) -> Optional[GraphIndex]:
    """The index ``place_graph_indices`` left for exactly this ``edge_index`` (same object, unmodified), in whichever
    node order the loader indexed the events in, or None.  ``ECForGraphTCN`` asks here first: a loader that keeps
    per-event indices (``io.ResidentDataset``) has already decided the node order of the batch."""
    for renumbered in (True, False):
        hit = _GI_CACHE.get((id(edge_index), renumbered))
        if hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version and hit[2] == n_nodes and hit[3].placed:
            return _join(hit[3])
    return None


def _carry_label_ok(y: Tensor, E: int, dev) -> bool:
    return y.dtype in (torch.bool, torch.uint8) and y.numel() == E and y.device == dev


def _carry_rows_ok(r: Tensor, E: int, dev) -> bool:
    return (r.device == dev and r.dtype == torch.float32 and r.dim() == 2 and tuple(r.shape) == (E, 4) and r.stride(1) == 1
            and r.stride(0) >= 4 and r.stride(0) % 4 == 0 and r.data_ptr() % 16 == 0)


def _carried(gi: GraphIndex, slot: str, t: Tensor):
    hit = getattr(gi, slot, None)
    if hit is not None and hit[0] == id(t) and hit[1] == t._version and hit[2]() is t:
        return hit[3]
    return None


def carried_label(gi: GraphIndex, y: Tensor):
    """uint8 ``[E]`` ``y[perm]`` if the build of ``gi`` carried exactly this ``y`` along, else None."""
    return _carried(gi, "_label_csr", y)


def carried_rows(gi: GraphIndex, rows: Tensor):
    """bf16 ``[E, 4]`` ``rows[perm]`` if the build of ``gi`` carried exactly this tensor along, else None."""
    return _carried(gi, "_rows_csr", rows)


_GI_CACHE_MAX = 4   # live entries: the EC graph, the cut graph and a prefetched batch or two


def _cache_put(edge_index: Tensor, n_nodes: int, gi: GraphIndex) -> None:
    """An index is 28 B/edge (1.8 GB at 64 M edges): entries whose edge list is gone are
    dropped on every insertion, and at most ``_GI_CACHE_MAX`` live ones are kept (oldest out)."""
    for k in [k for k, v in _GI_CACHE.items() if v[0]() is None]:
        _GI_CACHE.pop(k, None)
    while len(_GI_CACHE) >= _GI_CACHE_MAX:
        _GI_CACHE.pop(next(iter(_GI_CACHE)))
    _GI_CACHE[(id(edge_index), gi.order_sig is not None)] = (weakref.ref(edge_index), edge_index._version, n_nodes, gi)


def _join(gi: GraphIndex) -> GraphIndex:
    """First use of an index that was built on a side stream: the consumer's stream waits for
    the build and the arrays are handed over to it (allocator bookkeeping)."""
    if gi.ready is not None:
        cur = torch.cuda.current_stream(gi.perm.device)
        cur.wait_event(gi.ready)
        # every array the build allocated in the side stream's pool, the node order and the carried
        # label / edge-feature buffers included: without the hand-over the allocator may give their
        # blocks to the loader's next copy while queued consumer kernels still read them
        carried = [getattr(gi, slot)[3] for slot in ("_label_csr", "_rows_csr") if getattr(gi, slot, None) is not None]
        for t in (gi.perm, gi.tgt, gi.src, gi.rowptr_t, gi.rowptr_s, gi.spos, gi.spos_inv,
                  getattr(gi, "node_perm", None), getattr(gi, "node_rank", None), *carried):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(cur)
        gi.ready = None
    return gi


def prefetch_graph_index(edge_index: Tensor, n_nodes: int, stream, x: Optional[Tensor] = None,
                         batch: Optional[Tensor] = None) -> GraphIndex:
    """Build the index of ``edge_index`` on ``stream`` (a side stream: the loader's) while the
    current stream computes, and register it in the cache; ``graph_index()`` for the same
    tensor then joins it instead of building.  The index only depends on the input edge list,
    so a loader can have it ready one batch ahead (io.PrefetchLoader(build_index=True)).  ``x`` / ``batch``:
    the batch's node features and event ids - with them the index is built in the node order the edge
    classifier will ask for (locality.py), so that its lookup finds this build."""
    from . import locality
    col = None if x is None else locality.key_column(x)
    with torch.cuda.stream(stream):
        gi = graph_index(edge_index, n_nodes, cache=False, order_by=None if col is None else (x, col, batch))
        gi.ready = torch.cuda.Event()
        gi.ready.record(stream)
    _cache_put(edge_index, n_nodes, gi)
    return gi


# ---------------------------------------------------------------- plain helpers
def _segment_sum_raw(rows: Tensor, rowptr: Tensor, pos: Optional[Tensor], n_seg: int,
                     out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    lib = _capi.load()
    rows = _as_rows(rows)
    dim = rows.shape[1]
    if out is None:
        out = torch.empty(n_seg, dim, dtype=torch.float32, device=rows.device)
        accumulate = False
    _capi.check(lib.gnntrk_segment_sum(_p(rows), dim, _row_stride(rows), _p(rowptr), _p(pos),
                                       n_seg, _p(out), _row_stride(out), int(accumulate),
                                       _stream(rows)), lib)
    return out


def _permute_raw(x: Tensor, idx: Tensor, scatter: bool) -> Tensor:
    lib = _capi.load()
    x2 = _as_rows(x)
    m = int(idx.shape[0])  # gather: rows of the result; scatter: idx is a permutation of them
    if scatter and m != x2.shape[0]:
        raise ValueError("permute_rows(scatter): idx must be a permutation of the rows")
    out = torch.empty(m, x2.shape[1], dtype=torch.float32, device=x2.device)
    _capi.check(lib.gnntrk_permute_rows(_p(x2), x2.shape[1], _row_stride(x2), _p(idx), m,
                                        _p(out), _row_stride(out), int(scatter),
                                        _stream(x2)), lib)
    return out.view(-1) if x.dim() == 1 else out


def _axpby_raw(a: float, x: Tensor, b: float = 0.0, y: Optional[Tensor] = None,
               relu_mask: Optional[Tensor] = None) -> Tensor:
    lib = _capi.load()
    x = x.contiguous()
    y = None if y is None else y.contiguous()
    relu_mask = None if relu_mask is None else relu_mask.contiguous()
    out = torch.empty_like(x)
    _capi.check(lib.gnntrk_axpby(a, _p(x), b, _p(y), _p(relu_mask), _p(out), x.numel(),
                                 _stream(x)), lib)
    return out


class _SegmentSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, gi: GraphIndex, by: str):
        _capi.require_device(rows)
        ctx.gi, ctx.by = gi, by
        if by == "tgt":
            return _segment_sum_raw(rows, gi.rowptr_t, None, gi.n_nodes)
        if by == "src":
            return _segment_sum_raw(rows, gi.rowptr_s, gi.spos, gi.n_nodes)
        raise ValueError(by)

    @staticmethod
    def backward(ctx, g):
        # d/d rows[k] = g[node_of(k)]: a row gather by the CSR endpoint
        gi = ctx.gi
        idx = gi.tgt if ctx.by == "tgt" else gi.src
        return _permute_raw(g.contiguous(), idx, scatter=False), None, None


def segment_sum(rows: Tensor, gi: GraphIndex, by: str = "tgt") -> Tensor:
    """``out[n] = sum of rows[k] over CSR positions k whose target (source) is n``.
    ``rows`` must be in CSR order.  PyG ``aggr="add"`` (interaction_network.py:36)."""
    if rows.dtype == torch.bfloat16:
        if rows.dim() == 2 and rows.shape[1] > 16:  # wider than the bf16 kernel's four chunks: fp32 kernel
            return _SegmentSum.apply(rows.float(), gi, by).to(torch.bfloat16)
        from . import ops_bf16
        return ops_bf16.SegmentSum16.apply(rows, gi, by)
    return _SegmentSum.apply(rows, gi, by)


class _PermuteRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, scatter: bool):
        _capi.require_device(x)
        ctx.idx, ctx.scatter = idx, scatter
        return _permute_raw(x, idx, scatter)

    @staticmethod
    def backward(ctx, g):
        return _permute_raw(g.contiguous(), ctx.idx, not ctx.scatter), None, None


def permute_rows(x: Tensor, idx: Tensor, scatter: bool = False) -> Tensor:
    """gather: ``out[m] = x[idx[m]]``; scatter: ``out[idx[m]] = x[m]`` (idx a permutation)."""
    if x.dtype == torch.bfloat16:
        from . import ops_bf16
        return ops_bf16.PermuteRows16.apply(x, idx, scatter)
    return _PermuteRows.apply(x, idx, scatter)


class _Axpby(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, a: float, b: float):
        _capi.require_device(x, y)
        ctx.a, ctx.b = a, b
        return _axpby_raw(a, x, b, y)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return _axpby_raw(ctx.a, g), _axpby_raw(ctx.b, g), None, None


def axpby(a: float, x: Tensor, b: float, y: Tensor) -> Tensor:
    return _Axpby.apply(x, y, a, b)


# -------------------------------------------------------------------- fused MLP
@dataclasses.dataclass
class Seg:
    """One concat segment of a fused-MLP input.

    ``t``      source rows ``[R, dim]`` (or ``[R]``)
    ``idx``    int32 ``[M]`` row gather (None: row m of the op reads row m of ``t``)
    ``relu``   ReLU applied on load (resin.py:103-104)
    ``reduce`` how the backward folds per-row gradients onto ``t``:
               None (identity rows), "perm" (idx is a permutation), or
               ("tgt"|"src", GraphIndex) for node rows gathered by CSR endpoint.
    """

    t: Tensor
    idx: Optional[Tensor] = None
    relu: bool = False
    reduce: object = None


@dataclasses.dataclass
class _MlpSpec:
    n_seg: int
    n_layers: int
    has_bias: bool
    idx: list
    relu: list
    reduce: list
    epilogue: int
    ca: float
    cb: float
    out_idx: Optional[Tensor]
    out_rows: int
    n_rows: int


def _fill_mlp(weights: Sequence[Tensor], biases: Sequence[Optional[Tensor]]):
    n_layers = len(weights)
    hidden = weights[0].shape[0]
    in_dim = weights[0].shape[1]
    out_dim = weights[-1].shape[0]
    if n_layers not in (2, 3):
        raise NotImplementedError(
            f"fused MLP kernels cover L=2 and L=3 (the reference's uses); got L={n_layers}")
    for i, w in enumerate(weights):
        exp = (hidden if i < n_layers - 1 else out_dim, in_dim if i == 0 else hidden)
        if tuple(w.shape) != exp:
            raise ValueError(f"layer {i} weight has shape {tuple(w.shape)}, expected {exp}")
    # (the descriptor only; each storage mode's launcher checks its own limits - fp32: in <= 48,
    #  hidden <= 64; bf16: 16 input chunks, hidden + bias row <= 96, <= 128 with one k-step of inputs)
    if in_dim > _capi.MAX_IN_BF16 or hidden > _capi.MAX_HIDDEN_BF16 or out_dim > _capi.MAX_OUT_BF16:
        raise NotImplementedError(
            f"fused MLP kernel limits: in<={_capi.MAX_IN_BF16}, hidden<={_capi.MAX_HIDDEN_BF16}, "
            f"out<={_capi.MAX_OUT_BF16}; got in={in_dim}, hidden={hidden}, out={out_dim}")
    return _capi.make_mlp([_p(w) for w in weights], [_p(b) for b in biases], in_dim, hidden,
                          out_dim)


#: parameters marked by ``dist.FlatParameters`` own PERSISTENT fp32 gradient buffers (views of one
#: bucket): the reduction at the end of a backward launch then adds into them directly
#: (``gnntrk_mlp_bwd_args.accumulate_params``) and autograd gets ``None`` for those inputs, instead
#: of six small tensors per MLP and one ``add_`` kernel each.  Because ``AccumulateGrad`` then never
#: runs for such a parameter, the shortcut is only taken where that is known to be harmless:
#: INSIDE ``grad_sinks_armed()`` - which this package's own optimisation step puts around its plain
#: ``loss.backward()`` (``training.TrackingModule.backward_step``) - and for parameters without tensor
#: hooks / post-accumulate-grad hooks.  A ``loss.backward()`` or ``torch.autograd.grad`` issued by
#: anyone else (a Lightning loop, DDP's reducer hooks on the accumulators, a user's own hooks) gets
#: the ordinary autograd path: gradients are returned, hooks fire.  ``GNNTRK_GRAD_SINK=0`` turns the
#: shortcut off altogether.
_GRAD_SINK = os.environ.get("GNNTRK_GRAD_SINK", "1") != "0"
_SINKS_ARMED = 0


@contextlib.contextmanager
def grad_sinks_armed():
    """Around a plain ``tensor.backward()`` whose marked parameters (``dist.FlatParameters(grad_sink=True)``)
    may receive their gradients in place from the backward launches."""
    global _SINKS_ARMED
    _SINKS_ARMED += 1
    try:
        yield
    finally:
        _SINKS_ARMED -= 1


def _param_grad_sinks(weights, biases, need_w, need_b):
    """``(gW buffers, gb buffers)`` to accumulate into, or ``None``: only inside ``grad_sinks_armed()``, and
    every parameter of the launch that needs a gradient must be marked, free of hooks and hold a matching
    contiguous fp32 ``.grad``."""
    if not _GRAD_SINK or _SINKS_ARMED <= 0:
        return None
    gW, gb = [], []
    for plist, need, out in ((weights, need_w, gW), (biases, need_b, gb)):
        for p_, nd in zip(plist, need):
            if p_ is None:
                out.append(None)
                continue
            g = getattr(p_, "grad", None)
            if (not nd or not getattr(p_, "_gnntrk_grad_sink", False) or g is None or g.dtype != torch.float32
                    or g.shape != p_.shape or g.device != p_.device or not g.is_contiguous()
                    or getattr(p_, "_backward_hooks", None) or getattr(p_, "_post_accumulate_grad_hooks", None)):
                return None
            out.append(g)
    return gW, gb


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: _MlpSpec, *tensors):
        ns, nl = spec.n_seg, spec.n_layers
        segs = [_as_rows(t) for t in tensors[:ns]]
        weights = [w.contiguous() for w in tensors[ns:ns + nl]]
        biases = [None if b is None else b.contiguous() for b in tensors[ns + nl:ns + 2 * nl]]
        res = tensors[ns + 2 * nl]
        _capi.require_device(*segs, *weights)
        lib = _capi.load()
        a = _capi.MlpFwdArgs()
        a.mlp = _fill_mlp(weights, biases)
        if sum(s.shape[1] for s in segs) != a.mlp.in_dim:
            raise AssertionError(
                f"Expected feature dimension {a.mlp.in_dim}, got {sum(s.shape[1] for s in segs)}")
        a.n_seg, a.epilogue, a.n_rows = ns, spec.epilogue, spec.n_rows
        for j, s in enumerate(segs):
            a.seg[j] = _capi.Seg(_p(s), _p(spec.idx[j]), s.shape[1], _row_stride(s),
                                 int(spec.relu[j]), 0)
        a.ca, a.cb = spec.ca, spec.cb
        if spec.epilogue == _capi.EPI_RESIDUAL:
            res = _as_rows(res)
            a.res, a.res_stride = _p(res), _row_stride(res)
        out = torch.empty(spec.out_rows, a.mlp.out_dim, dtype=torch.float32,
                          device=segs[0].device)
        a.out, a.out_stride, a.out_idx = _p(out), _row_stride(out), _p(spec.out_idx)
        M = spec.n_rows
        nbytes = M * (sum(4 * s.shape[1] + (4 if spec.idx[j] is not None else 0)
                          for j, s in enumerate(segs)) + 4 * a.mlp.out_dim
                      + (4 if spec.out_idx is not None else 0)
                      + (4 * a.mlp.out_dim if spec.epilogue == _capi.EPI_RESIDUAL else 0))
        with _timed(out, kernel_key(lib, a, False),
                    _mlp_flops_per_row(a.mlp) * M, nbytes, M):
            _capi.check(lib.gnntrk_mlp_forward(C.byref(a), _stream(out)), lib)
        ctx.spec = spec
        ctx.save_for_backward(*segs, *weights, *[b for b in biases if b is not None])
        ctx.bias_mask = [b is not None for b in biases]
        return out

    @staticmethod
    def backward(ctx, g_out):
        spec: _MlpSpec = ctx.spec
        ns, nl = spec.n_seg, spec.n_layers
        saved = ctx.saved_tensors
        segs, weights = list(saved[:ns]), list(saved[ns:ns + nl])
        bl = list(saved[ns + nl:])
        biases = [bl.pop(0) if m else None for m in ctx.bias_mask]
        lib = _capi.load()
        g_out = _as_rows(g_out.contiguous())
        dev = g_out.device
        a = _capi.MlpBwdArgs()
        a.mlp = _fill_mlp(weights, biases)
        a.n_seg, a.epilogue, a.n_rows = ns, spec.epilogue, spec.n_rows
        a.ca, a.cb = spec.ca, spec.cb
        for j, s in enumerate(segs):
            a.seg[j] = _capi.Seg(_p(s), _p(spec.idx[j]), s.shape[1], _row_stride(s),
                                 int(spec.relu[j]), 0)
        a.n_gout = 1
        a.gout[0] = _capi.GTerm(_p(g_out), _p(spec.out_idx), _row_stride(g_out), 0)

        need = ctx.needs_input_grad  # [spec, segs..., W..., b..., res]
        M = spec.n_rows
        seg_grads: list[Optional[Tensor]] = [None] * ns
        row_tmp: list[Optional[Tensor]] = [None] * ns
        for j, s in enumerate(segs):
            if not need[1 + j]:
                continue
            red = spec.reduce[j]
            if spec.idx[j] is None:
                gj = torch.empty(s.shape[0], s.shape[1], dtype=torch.float32, device=dev)
                if s.shape[0] != M:
                    gj.zero_()
                seg_grads[j] = gj
                a.gseg[j] = _capi.GSeg(_p(gj), None, _row_stride(gj), 0)
            else:
                if red is None:
                    raise RuntimeError("gathered segment requires a `reduce` rule for backward")
                tmp = torch.empty(M, s.shape[1], dtype=torch.float32, device=dev)
                row_tmp[j] = tmp
                a.gseg[j] = _capi.GSeg(_p(tmp), None, _row_stride(tmp), 0)

        want_dw = any(need[1 + ns:1 + ns + 2 * nl])
        gW = [None] * nl
        gb = [None] * nl
        ws = None
        sinks = None
        if want_dw:
            sinks = _param_grad_sinks(weights, biases, need[1 + ns:1 + ns + nl],
                                      [need[1 + ns + nl + i] or biases[i] is None for i in range(nl)])
            if sinks is not None:
                gW, gb = sinks
            else:
                gW = [torch.empty_like(w) for w in weights]
                gb = [None if b is None else torch.empty_like(b) for b in biases]
            for i in range(nl):
                a.gW[i] = _p(gW[i])
                a.gb[i] = _p(gb[i])
            ws = _ws(lib.gnntrk_mlp_backward_workspace_bytes(C.byref(a.mlp)), g_out)
        a.accumulate_params = 1 if sinks is not None else 0
        a.debug_flags = int(__import__("os").environ.get("GNNTRK_DEBUG_FLAGS", "0"))
        nbytes = M * (sum(4 * s.shape[1] + (4 if spec.idx[j] is not None else 0)
                          + (4 * s.shape[1] if need[1 + j] else 0)
                          for j, s in enumerate(segs)) + 4 * a.mlp.out_dim
                      + (4 if spec.out_idx is not None else 0))
        if __import__("os").environ.get("GNNTRK_DEBUG_PTRS"):
            def rng(t):
                return "None" if t is None else f"[{t.data_ptr():#x},{t.data_ptr() + t.numel() * t.element_size():#x})"
            print("bwd ptrs: g_out", rng(g_out), "ws", rng(ws), "segs", [rng(s) for s in segs],
                  "W", [rng(w) for w in weights], "gW", [rng(w) for w in gW], "gb", [rng(b) for b in gb],
                  "gseg", [rng(t) for t in seg_grads], "tmp", [rng(t) for t in row_tmp],
                  "M", M, "dims", a.mlp.in_dim, a.mlp.hidden, a.mlp.out_dim, flush=True)
        with _timed(g_out, kernel_key(lib, a, True),
                    3 * _mlp_flops_per_row(a.mlp) * M, nbytes, M):
            _capi.check(lib.gnntrk_mlp_backward(C.byref(a), _p(ws),
                                                0 if ws is None else ws.numel(),
                                                _stream(g_out)), lib)

        # fold gathered row gradients onto their source rows (deterministic CSR sums)
        for j, s in enumerate(segs):
            if row_tmp[j] is None:
                continue
            if spec.reduce[j] == "perm":  # idx is a permutation of the source rows
                if s.shape[0] != M:
                    raise RuntimeError("'perm' segments must cover all source rows")
                seg_grads[j] = _permute_raw(row_tmp[j], spec.idx[j], scatter=True)
                continue
            by, gi = spec.reduce[j]
            rowptr, pos = (gi.rowptr_t, None) if by == "tgt" else (gi.rowptr_s, gi.spos)
            seg_grads[j] = _segment_sum_raw(row_tmp[j], rowptr, pos, s.shape[0])

        g_res = None
        if spec.epilogue == _capi.EPI_RESIDUAL and need[1 + ns + 2 * nl]:
            g_dense = g_out if spec.out_idx is None else _permute_raw(g_out, spec.out_idx, False)
            g_res = _axpby_raw(spec.ca, g_dense.contiguous())
        outs = [None, *seg_grads]
        if sinks is not None:   # (already added into the parameters' gradient buffers)
            outs += [None] * (2 * nl)
        else:
            outs += [gW[i] if need[1 + ns + i] else None for i in range(nl)]
            outs += [gb[i] if need[1 + ns + nl + i] else None for i in range(nl)]
        outs.append(g_res)
        return tuple(outs)


def fused_mlp(segs: Sequence[Seg], weights: Sequence[Tensor],
              biases: Sequence[Optional[Tensor]], *, n_rows: Optional[int] = None,
              epilogue: int = _capi.EPI_NONE, ca: float = 0.0, cb: float = 1.0,
              res: Optional[Tensor] = None, out_idx: Optional[Tensor] = None,
              out_rows: Optional[int] = None) -> Tensor:
    """``epilogue( MLP( concat_j act_j(gather_j(seg_j)) ) )`` -> ``[out_rows, out_dim]``."""
    if not 1 <= len(segs) <= _capi.MAX_SEGS:
        raise ValueError(f"1..{_capi.MAX_SEGS} segments supported, got {len(segs)}")
    if n_rows is None:
        s0 = segs[0]
        n_rows = int(s0.idx.shape[0]) if s0.idx is not None else int(s0.t.shape[0])
    for s in segs:
        if s.t.dim() == 1:
            raise ValueError("fused_mlp segments must be 2-D [rows, dim]")
    bf16 = segs[0].t.dtype == torch.bfloat16
    if not _fused_supported(segs, weights, biases, bf16, epilogue):
        if not bf16 and out_idx is None and _wide_kernel_supported(segs, weights, epilogue):
            # fp32 beyond the register-resident kernels (in <= 128, hidden <= 128, out <= 48): the LDS-staged
            # kernels of csrc/mlp_wide.hip - one forward, one backward launch
            spec = _MlpSpec(len(segs), len(weights), any(b is not None for b in biases),
                            [s.idx for s in segs], [s.relu for s in segs], [s.reduce for s in segs],
                            epilogue, float(ca), float(cb), None, int(n_rows), int(n_rows))
            # (inside forward() grad mode is off and needs_input_grad is True for trainable parameters even under
            #  no_grad(): the caller's grad mode travels on the spec - inference does not store the activations)
            spec.grad_on = torch.is_grad_enabled()
            return _FusedMLPWide.apply(spec, *[s.t for s in segs], *weights, *biases, res)
        return _wide_mlp(segs, weights, biases, n_rows=int(n_rows), epilogue=epilogue, ca=float(ca), cb=float(cb),
                         res=res, out_idx=out_idx, out_rows=int(out_rows if out_rows is not None else n_rows))
    spec = _MlpSpec(len(segs), len(weights), any(b is not None for b in biases),
                    [s.idx for s in segs], [s.relu for s in segs], [s.reduce for s in segs],
                    epilogue, float(ca), float(cb), out_idx,
                    int(out_rows if out_rows is not None else n_rows), int(n_rows))
    if bf16:  # bf16-storage path (ops_bf16.py)
        from . import ops_bf16
        return ops_bf16.FusedMLP16.apply(spec, *[s.t for s in segs], *weights, *biases, res)
    return _FusedMLP.apply(spec, *[s.t for s in segs], *weights, *biases, res)


# ------------------------------------------------ MLPs beyond the fused kernels' shapes
#: four-feature input chunks the bf16 kernels take (wide inputs: with three hidden tiles only)
_BF16_MAX_CHUNKS = 32


def _fused_supported(segs: Sequence[Seg], weights: Sequence[Tensor], biases, bf16: bool, epilogue=None) -> bool:
    """The shapes the register-resident fused kernels hold (include/gnntrk.h: L in {2, 3}; at
    most sixteen 4-feature input chunks; fp32: in <= 48, hidden <= 64, out <= 16; bf16 storage:
    hidden (+ the bias row, which 64 and 128 do without) <= 96 - <= 128 with at most eight input chunks -,
    out <= 16; with hidden + bias row in 33 .. 48 also 32 input chunks and out <= 48, NONE / RESIDUAL)."""
    L = len(weights)
    if L not in (2, 3):
        return False
    hidden, out_dim = int(weights[0].shape[0]), int(weights[-1].shape[0])
    in_dim = sum(int(s.t.shape[1]) if s.t.dim() == 2 else 1 for s in segs)
    chunks = sum((int(s.t.shape[1]) + 3) // 4 for s in segs if s.t.dim() == 2)
    if out_dim > (_capi.MAX_OUT_BF16 if bf16 else _capi.MAX_OUT):
        return False
    if bf16:
        has_bias = any(b is not None for b in biases)
        spare = any(int(s.t.shape[1]) % 4 for s in segs if s.t.dim() == 2)   # a pad slot carries the ones column
        n_ch = chunks + (1 if has_bias and not spare else 0)
        hid_bias = any(b is not None for b in list(biases)[1:])   # a bias after the first layer
        # (the constant-one hidden row; 64 and - with one k-step of inputs - 128 do without it: tile_bf16.h)
        rows = hidden + (1 if hid_bias and not (hidden == 64 or (hidden == 128 and n_ch <= 8)) else 0)
        if out_dim > 16 or n_ch > 16:   # output tiles / wide inputs: the three-hidden-tile instantiations only
            # (their backward takes the NONE / RESIDUAL epilogues)
            return (n_ch <= _BF16_MAX_CHUNKS and 33 <= rows <= 48
                    and epilogue in (None, _capi.EPI_NONE, _capi.EPI_RESIDUAL))
        return n_ch <= 16 and rows <= (128 if n_ch <= 8 else 96)
    return in_dim <= _capi.MAX_IN and hidden <= _capi.MAX_HIDDEN and chunks <= 16


#: the wide fp32 kernels on / off (off: library GEMMs for those shapes, as before round 4)
_WIDE_KERNEL = os.environ.get("GNNTRK_WIDE_KERNEL", "1") != "0"


def _wide_kernel_supported(segs: Sequence[Seg], weights: Sequence[Tensor], epilogue) -> bool:
    """Shapes ``gnntrk_mlp_forward_wide`` / ``_backward_wide`` take (include/gnntrk.h)."""
    if not _WIDE_KERNEL or len(weights) not in (2, 3):
        return False
    if any(s.t.dtype != torch.float32 for s in segs) or any(w.dtype != torch.float32 for w in weights):
        return False
    hidden, out_dim = int(weights[0].shape[0]), int(weights[-1].shape[0])
    in_dim = sum(int(s.t.shape[1]) for s in segs)
    return (in_dim <= _capi.WIDE_MAX_IN and hidden <= _capi.WIDE_MAX_HIDDEN and out_dim <= _capi.WIDE_MAX_OUT
            and epilogue in (None, _capi.EPI_NONE, _capi.EPI_RELU, _capi.EPI_RESIDUAL))


def _wide_forward(ctx, spec: "_MlpSpec", tensors, needs_grad: bool):
    """Forward launch of the wide fp32 kernels for ``_FusedMLPWide`` / ``_FusedINEdgeWide``: fills ``ctx`` for
    ``_wide_backward`` and returns the output rows."""
    ns, nl = spec.n_seg, spec.n_layers
    segs = [_as_rows(t) for t in tensors[:ns]]
    weights = [w.contiguous() for w in tensors[ns:ns + nl]]
    biases = [None if b is None else b.contiguous() for b in tensors[ns + nl:ns + 2 * nl]]
    res = tensors[ns + 2 * nl] if len(tensors) > ns + 2 * nl else None
    _capi.require_device(*segs, *weights)
    lib = _capi.load()
    a = _capi.MlpFwdArgs()
    a.mlp = _fill_mlp(weights, biases)
    if sum(s.shape[1] for s in segs) != a.mlp.in_dim:
        raise AssertionError(
            f"Expected feature dimension {a.mlp.in_dim}, got {sum(s.shape[1] for s in segs)}")
    a.n_seg, a.epilogue, a.n_rows = ns, spec.epilogue, spec.n_rows
    for j, s in enumerate(segs):
        a.seg[j] = _capi.Seg(_p(s), _p(spec.idx[j]), s.shape[1], _row_stride(s), int(spec.relu[j]), 0)
    a.ca, a.cb = spec.ca, spec.cb
    if spec.epilogue == _capi.EPI_RESIDUAL:
        res = _as_rows(res)
        a.res, a.res_stride = _p(res), _row_stride(res)
    M = spec.n_rows
    dev = segs[0].device
    out = torch.empty(M, a.mlp.out_dim, dtype=torch.float32, device=dev)
    a.out, a.out_stride = _p(out), _row_stride(out)
    acts = None
    if needs_grad and M > 0:
        hp = int(lib.gnntrk_mlp_wide_hidden_pad(a.mlp.hidden))
        acts = torch.empty(nl - 1, M, hp, dtype=torch.float32, device=dev)
    ws = _ws(lib.gnntrk_mlp_wide_forward_workspace_bytes(C.byref(a.mlp)), out)
    _capi.check(lib.gnntrk_mlp_forward_wide(C.byref(a), _p(acts), _p(ws), ws.numel(), _stream(out)), lib)
    ctx.spec = spec
    ctx.save_for_backward(acts, out if spec.epilogue == _capi.EPI_RELU else None, *segs, *weights,
                          *[b for b in biases if b is not None])
    ctx.bias_mask = [b is not None for b in biases]
    return out


def _wide_backward(ctx, gterms, need):
    """Backward launch of the wide fp32 kernels.  ``gterms``: up to three ``(rows, index or None)`` upstream
    gradient terms, summed per row inside the kernel (include/gnntrk.h: gout); ``need``:
    ``[spec, segs..., W..., b..., res]``.  Returns the gradients in that order."""
    spec: _MlpSpec = ctx.spec
    ns, nl = spec.n_seg, spec.n_layers
    saved = ctx.saved_tensors
    acts, out = saved[0], saved[1]
    segs, weights = list(saved[2:2 + ns]), list(saved[2 + ns:2 + ns + nl])
    bl = list(saved[2 + ns + nl:])
    biases = [bl.pop(0) if m else None for m in ctx.bias_mask]
    lib = _capi.load()
    gterms = [(_as_rows(g.contiguous()), idx) for g, idx in gterms]
    dev = gterms[0][0].device
    a = _capi.MlpBwdArgs()
    a.mlp = _fill_mlp(weights, biases)
    a.n_seg, a.epilogue, a.n_rows = ns, spec.epilogue, spec.n_rows
    a.ca, a.cb = spec.ca, spec.cb
    for j, s in enumerate(segs):
        a.seg[j] = _capi.Seg(_p(s), _p(spec.idx[j]), s.shape[1], _row_stride(s), int(spec.relu[j]), 0)
    a.n_gout = len(gterms)
    for t, (g, idx) in enumerate(gterms):
        a.gout[t] = _capi.GTerm(_p(g), _p(idx), _row_stride(g), 0)
    M = spec.n_rows
    seg_grads: list[Optional[Tensor]] = [None] * ns
    row_tmp: list[Optional[Tensor]] = [None] * ns
    for j, s in enumerate(segs):
        if not need[1 + j]:
            continue
        if spec.idx[j] is None:
            if s.shape[0] != M:
                raise RuntimeError("identity segments must have n_rows rows")
            gj = torch.empty(M, s.shape[1], dtype=torch.float32, device=dev)
            seg_grads[j] = gj
            a.gseg[j] = _capi.GSeg(_p(gj), None, _row_stride(gj), 0)
        else:
            if spec.reduce[j] is None:
                raise RuntimeError("gathered segment requires a `reduce` rule for backward")
            tmp = torch.empty(M, s.shape[1], dtype=torch.float32, device=dev)
            row_tmp[j] = tmp
            a.gseg[j] = _capi.GSeg(_p(tmp), None, _row_stride(tmp), 0)
    want_dw = any(need[1 + ns:1 + ns + 2 * nl])
    gW, gb = [None] * nl, [None] * nl
    sinks = None
    if want_dw:
        sinks = _param_grad_sinks(weights, biases, need[1 + ns:1 + ns + nl],
                                  [need[1 + ns + nl + i] or biases[i] is None for i in range(nl)])
        if sinks is not None:
            gW, gb = sinks
        else:
            gW = [torch.empty_like(w) for w in weights]
            gb = [None if b is None else torch.empty_like(b) for b in biases]
        for i in range(nl):
            a.gW[i] = _p(gW[i])
            a.gb[i] = _p(gb[i])
    a.accumulate_params = 1 if sinks is not None else 0
    g0 = gterms[0][0]
    ws = _ws(lib.gnntrk_mlp_wide_backward_workspace_bytes(C.byref(a.mlp), M), g0)
    _capi.check(lib.gnntrk_mlp_backward_wide(C.byref(a), _p(acts), _p(out),
                                             0 if out is None else _row_stride(out), _p(ws), ws.numel(),
                                             _stream(g0)), lib)
    for j, s in enumerate(segs):
        if row_tmp[j] is None:
            continue
        if spec.reduce[j] == "perm":
            if s.shape[0] != M:
                raise RuntimeError("'perm' segments must cover all source rows")
            seg_grads[j] = _permute_raw(row_tmp[j], spec.idx[j], scatter=True)
            continue
        by, gi = spec.reduce[j]
        rowptr, pos = (gi.rowptr_t, None) if by == "tgt" else (gi.rowptr_s, gi.spos)
        seg_grads[j] = _segment_sum_raw(row_tmp[j], rowptr, pos, s.shape[0])
    g_res = None
    if spec.epilogue == _capi.EPI_RESIDUAL and need[1 + ns + 2 * nl]:
        g_res = _axpby_raw(spec.ca, _sum_terms(gterms))
    outs = [None, *seg_grads]
    if sinks is not None:
        outs += [None] * (2 * nl)
    else:
        outs += [gW[i] if need[1 + ns + i] else None for i in range(nl)]
        outs += [gb[i] if need[1 + ns + nl + i] else None for i in range(nl)]
    outs.append(g_res)
    return tuple(outs)


def _sum_terms(gterms) -> Tensor:
    """The upstream gradient the kernel sums per row, materialised (only the residual epilogue's pass-through
    gradient needs it, and that one has a single un-gathered term)."""
    total = None
    for g, idx in gterms:
        rows = g if idx is None else g.index_select(0, idx.long())
        total = rows if total is None else total + rows
    return total.contiguous()


class _FusedMLPWide(torch.autograd.Function):
    """``_FusedMLP`` on the wide fp32 kernels (csrc/mlp_wide.hip): same spec, same fold rules for gathered
    segments; the forward keeps the hidden layers' pre-activations for the backward."""

    @staticmethod
    def forward(ctx, spec: _MlpSpec, *tensors):
        return _wide_forward(ctx, spec, tensors, getattr(spec, "grad_on", True) and any(ctx.needs_input_grad))

    @staticmethod
    def backward(ctx, g_out):
        return _wide_backward(ctx, [(g_out, None)], ctx.needs_input_grad)


class _FusedINEdgeWide(torch.autograd.Function):
    """Relational model + sum aggregation of one interaction-network layer (interaction_network.py:67-89) as
    ONE autograd node on the wide fp32 kernels: ``(e~, aggr) = f(segments, params)``.  The backward hands the
    kernel both upstream terms - ``g_e~[k] + g_aggr[tgt[k]]`` - so the aggregation's gradient is never
    gathered into an edge-sized tensor (the fp32 twin of ``ops_bf16.FusedINEdge16``)."""

    @staticmethod
    def forward(ctx, spec: _MlpSpec, gi: GraphIndex, *tensors):
        e_tilde = _wide_forward(ctx, spec, tensors, getattr(spec, "grad_on", True) and any(ctx.needs_input_grad))
        ctx.gi = gi
        ctx.set_materialize_grads(False)
        return e_tilde, _segment_sum_raw(e_tilde, gi.rowptr_t, None, gi.n_nodes)

    @staticmethod
    def backward(ctx, g_et, g_aggr):
        gterms = []
        if g_et is not None:
            gterms.append((g_et, None))
        if g_aggr is not None:
            gterms.append((g_aggr, ctx.gi.tgt))
        need = ctx.needs_input_grad  # [spec, gi, segs..., W..., b...]
        if not gterms:
            return (None,) * len(need)
        outs = _wide_backward(ctx, gterms, (need[0],) + tuple(need[2:]) + (False,))
        return (None, None, *outs[1:-1])


def in_edge_wide(segs, weights, biases, gi: GraphIndex, n_rows: int):
    """``(e~, aggr)`` of one interaction-network layer in fp32 on the wide kernels (see _FusedINEdgeWide)."""
    spec = _MlpSpec(len(segs), len(weights), any(b is not None for b in biases),
                    [s.idx for s in segs], [s.relu for s in segs], [s.reduce for s in segs],
                    _capi.EPI_NONE, 0.0, 1.0, None, int(n_rows), int(n_rows))
    spec.grad_on = torch.is_grad_enabled()
    return _FusedINEdgeWide.apply(spec, gi, *[s.t for s in segs], *weights, *biases)


class _GatherRows(torch.autograd.Function):
    """``out[m] = t[idx[m]]``; the backward folds the row gradients onto the source rows with the
    deterministic CSR segment sums of the graph index (``reduce`` as in ``Seg``)."""

    @staticmethod
    def forward(ctx, t, idx, reduce):
        ctx.idx, ctx.reduce, ctx.n, ctx.dt = idx, reduce, int(t.shape[0]), t.dtype
        return t.index_select(0, idx.long())

    @staticmethod
    def backward(ctx, g):
        g32 = g.to(torch.float32).contiguous()
        if ctx.reduce == "perm":
            out = _permute_raw(g32, ctx.idx, scatter=True)
        elif ctx.reduce is None:
            raise RuntimeError("gathered segment requires a `reduce` rule for backward")
        else:
            by, gi = ctx.reduce
            rowptr, pos = (gi.rowptr_t, None) if by == "tgt" else (gi.rowptr_s, gi.spos)
            out = _segment_sum_raw(g32, rowptr, pos, ctx.n)
        return out.to(ctx.dt), None, None


_WIDE_WARNED: set = set()


def _wide_mlp(segs: Sequence[Seg], weights, biases, *, n_rows: int, epilogue: int, ca: float, cb: float,
              res: Optional[Tensor], out_idx: Optional[Tensor], out_rows: int) -> Tensor:
    """The same operator for shapes the fused kernels do not hold (e.g. ``GraphConstructionResIN``
    at its default ``hidden_dim=40``: a 120-wide relational input; 128-wide models; L = 1 or
    L > 3): row gathers, ``cat``, a chain of library GEMMs (hipBLASLt through
    ``torch.nn.functional.linear``), the epilogue as torch ops.  Gathered node rows still get
    their gradients through the deterministic segment sums.  Slower than the fused kernels (the
    layer outputs travel through HBM) - a one-time notice says so."""
    import torch.nn.functional as F

    bf16 = segs[0].t.dtype == torch.bfloat16
    shape = (sum(int(s.t.shape[1]) for s in segs), *[int(w.shape[0]) for w in weights])
    if shape not in _WIDE_WARNED:
        _WIDE_WARNED.add(shape)
        import logging
        logging.getLogger("gnn_tracking_amd").info(
            "MLP %s is outside the fused kernels' shapes; running it as library GEMMs", "->".join(map(str, shape)))
    _capi.require_device(*[s.t for s in segs])
    cols = []
    for s_ in segs:
        t = s_.t
        if s_.relu:
            t = torch.relu(t)
        if s_.idx is not None:
            t = _GatherRows.apply(t, s_.idx, s_.reduce)
        cols.append(t)
    x = cols[0] if len(cols) == 1 else torch.cat(cols, dim=1)
    if x.shape[0] != n_rows:
        raise ValueError(f"segments have {x.shape[0]} rows, expected {n_rows}")
    dt = torch.bfloat16 if bf16 else torch.float32
    L = len(weights)
    for i, (w, b) in enumerate(zip(weights, biases)):
        x = F.linear(x.to(dt), w.to(dt), None if b is None else b.to(dt))
        if i < L - 1:
            x = torch.relu(x)
    if epilogue == _capi.EPI_RELU:
        x = torch.relu(x)
    elif epilogue == _capi.EPI_RESIDUAL:
        x = ca * res.to(x.dtype) + cb * x
    elif epilogue == _capi.EPI_SIGMOID:
        x = ca + cb * torch.sigmoid(x.float())   # fp32 in both storage modes
    if out_idx is not None:
        if out_rows != n_rows:
            raise ValueError("out_idx must be a permutation of the rows")
        x = permute_rows(x.contiguous(), out_idx, scatter=True)
    if bf16 and x.dtype == torch.bfloat16:
        # hand the rows on in the padded layout the bf16 kernels expect
        d, pad = int(x.shape[1]), (-int(x.shape[1])) % 4
        x = F.pad(x, (0, pad))[:, :d] if pad else x.contiguous()
    return x


# ------------------------------------------------------------------- kNN graphs
#: bit 0: pruned search below its row threshold too; bit 1: brute force only (tests, measurements)
_KNN_FLAGS = int(os.environ.get("GNNTRK_KNN_FLAGS", "0"))


def _knn_search(lib, x: Tensor, k: int, r: float, seg_ptr: Optional[Tensor], nbr: Tensor, cnt: Tensor, st) -> None:
    """``gnntrk_knn_search_ws``: the neighbour search with the workspace of its pruned form
    (identical output to the brute-force entry points; the library picks the form)."""
    n, dim = int(x.shape[0]), int(x.shape[1])
    nb = int(lib.gnntrk_knn_workspace_bytes(n, dim, k))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
    n_seg = int(seg_ptr.numel()) - 1 if seg_ptr is not None else 0
    _capi.check(lib.gnntrk_knn_search_ws(_p(x), n, dim, _row_stride(x), k, r,
                                         _p(seg_ptr) if seg_ptr is not None else None, n_seg, _p(nbr), _p(cnt),
                                         _p(ws) if ws is not None else None, nb, _KNN_FLAGS, st), lib)


def knn_graph(x: Tensor, k: int, max_radius: Optional[float] = None, seg_ptr: Optional[Tensor] = None) -> Tensor:
    """``knn_with_max_radius`` (models/graph_construction.py:222-237): int64 ``[2, M]``
    edge index, row 0 = neighbour (source), row 1 = query (target), grouped by query,
    ascending distance, self excluded; with ``max_radius`` only ``||x_j - x_i|| < r``.
    ``seg_ptr`` (int64 ``[S + 1]`` row offsets of the events of a collated batch): neighbours
    are searched inside the query's own event - torch_cluster's ``batch`` argument, all events
    in one launch."""
    _capi.require_device(x)
    lib = _capi.load()
    if x.dim() != 2:
        raise ValueError("knn_graph: x must be [N, D]")
    x = _as_rows(x.detach())
    n, dim = int(x.shape[0]), int(x.shape[1])
    dev = x.device
    if n <= 1:
        return torch.empty(2, 0, dtype=torch.int64, device=dev)
    kk = min(int(k), n - 1)
    nbr = torch.empty(n * kk, dtype=torch.int32, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    st = _stream(x)
    r = float(max_radius) if max_radius is not None else -1.0
    sp = seg_ptr.to(device=dev, dtype=torch.int64).contiguous() if seg_ptr is not None else None
    _knn_search(lib, x, kk, r, sp, nbr, cnt, st)
    off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    _capi.check(lib.gnntrk_knn_emit(_p(nbr), _p(cnt), n, kk, _p(off), None, 0, st), lib)
    m = int(off[n].item())  # the one host sync: the output size is data dependent
    ei = torch.empty(2, m, dtype=torch.int64, device=dev)
    _capi.check(lib.gnntrk_knn_emit(_p(nbr), _p(cnt), n, kk, _p(off), _p(ei), m, st), lib)
    return ei


def knn_kth_neighbor(x: Tensor, k: int, max_radius: Optional[float] = None) -> Tensor:
    """int32 ``[N]``: for every row the index of its ``k``-th nearest neighbour (inside
    ``max_radius``), -1 where it has fewer - the last column of one ``gnntrk_knn_search``.  Used
    for the neighbour cap of ``CondensationLossRG``'s radius graph."""
    _capi.require_device(x)
    lib = _capi.load()
    x = _as_rows(x.detach().to(torch.float32))
    n, dim = int(x.shape[0]), int(x.shape[1])
    if n - 1 < k:   # fewer candidates than the cap: it never binds
        return torch.full((n,), -1, dtype=torch.int32, device=x.device)
    nbr = torch.empty(n * k, dtype=torch.int32, device=x.device)
    cnt = torch.empty(n, dtype=torch.int32, device=x.device)
    r = float(max_radius) if max_radius is not None else -1.0
    _knn_search(lib, x, int(k), r, None, nbr, cnt, _stream(x))
    last = nbr.view(n, k)[:, k - 1]
    return torch.where(cnt >= k, last, torch.full_like(last, -1)).contiguous()


def max_radius_count(x: Tensor, radius: float) -> Tensor:
    """int32 scalar tensor (on the device): the largest number of OTHER rows of ``x`` within ``radius`` of a row - one
    count pass of the pruned radius graph (``gnntrk_radius_count_ws``: fp64 distances, ``d2 <= r^2``).  What decides
    whether the neighbour cap of ``CondensationLossRG``'s radius graph can bind at all."""
    _capi.require_device(x)
    lib = _capi.load()
    x = _as_rows(x.detach().to(torch.float32))
    n, dim = int(x.shape[0]), int(x.shape[1])
    if n < 2:
        return torch.zeros((), dtype=torch.int32, device=x.device)
    cnt = torch.empty(n, dtype=torch.int32, device=x.device)
    off = torch.empty(n + 1, dtype=torch.int64, device=x.device)
    nb = int(lib.gnntrk_radius_points_workspace_bytes(n, dim))
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device) if nb else None
    _capi.check(lib.gnntrk_radius_count_ws(_p(x), n, dim, _row_stride(x), float(radius), _p(cnt), _p(off),
                                           _p(ws) if ws is not None else None, nb, 0, _stream(x)), lib)
    return cnt.max() - 1   # (every row is its own neighbour in that graph)


def knn_scan(x: Tensor, ks: Sequence[int], max_radius: Optional[float] = None) -> dict:
    """``{k: knn_with_max_radius(x, k, max_radius) for k in ks}`` from ONE neighbour search at
    ``max(ks)`` (the k-scan of graph_construction/k_scanner.py:203-285 searches once per k):
    the ``k`` nearest are a prefix of every query's sorted neighbour list."""
    _capi.require_device(x)
    lib = _capi.load()
    if x.dim() != 2:
        raise ValueError("knn_scan: x must be [N, D]")
    ks = [int(k) for k in ks]
    if not ks or min(ks) < 1:
        raise ValueError("knn_scan: ks must be positive")
    x = _as_rows(x.detach())
    n, dim = int(x.shape[0]), int(x.shape[1])
    dev = x.device
    if n <= 1:
        return {k: torch.empty(2, 0, dtype=torch.int64, device=dev) for k in ks}
    kmax = min(max(ks), n - 1)
    nbr = torch.empty(n * kmax, dtype=torch.int32, device=dev)
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    st = _stream(x)
    r = float(max_radius) if max_radius is not None else -1.0
    _knn_search(lib, x, kmax, r, None, nbr, cnt, st)
    offs = {}
    for k in ks:  # all offset scans first, then ONE host read of the edge counts
        off = torch.empty(n + 1, dtype=torch.int64, device=dev)
        _capi.check(lib.gnntrk_knn_emit_prefix(_p(nbr), _p(cnt), n, kmax, min(k, kmax), _p(off), None, 0, st), lib)
        offs[k] = off
    totals = torch.stack([offs[k][n] for k in ks]).tolist()
    out = {}
    for k, m in zip(ks, totals):
        ei = torch.empty(2, int(m), dtype=torch.int64, device=dev)
        _capi.check(lib.gnntrk_knn_emit_prefix(_p(nbr), _p(cnt), n, kmax, min(k, kmax), _p(offs[k]), _p(ei),
                                               int(m), st), lib)
        out[k] = ei
    return out


def edge_labels(particle_id: Tensor, edge_index: Tensor) -> Tensor:
    """``(pid[e0] == pid[e1]) & (pid[e0] > 0)`` as int64 (graph_construction.py:365-367)."""
    _capi.require_device(particle_id, edge_index)
    lib = _capi.load()
    pid = particle_id.to(torch.int64).contiguous()
    ei = edge_index.contiguous()
    m = int(ei.shape[1])
    y = torch.empty(m, dtype=torch.int64, device=ei.device)
    _capi.check(lib.gnntrk_edge_labels(_p(pid), _p(ei), m, _p(y), _stream(ei)), lib)
    return y


class _EdgeFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, edge_index):
        lib = _capi.load()
        x2 = _as_rows(x.detach())
        ei = edge_index.contiguous()
        m, f = int(ei.shape[1]), int(x2.shape[1])
        out = torch.empty(m, 2 * f, dtype=torch.float32, device=x2.device)
        _capi.check(lib.gnntrk_edge_features(_p(x2), f, _row_stride(x2), _p(ei), m, _p(out),
                                             _stream(x2)), lib)
        ctx.ei, ctx.n, ctx.f = edge_index, int(x2.shape[0]), f
        return out

    @staticmethod
    def backward(ctx, g):
        # d/dx[n] = sum over edges with e0 = n of (g_diff + g_sum) + over edges with e1 = n of
        # (g_sum - g_diff): two deterministic segment sums over the graph index of the edge list
        f, n = ctx.f, ctx.n
        gi = graph_index(ctx.ei, n)
        g0 = _permute_raw((g[:, :f] + g[:, f:]).contiguous(), gi.perm, scatter=False)
        g1 = _permute_raw((g[:, f:] - g[:, :f]).contiguous(), gi.perm, scatter=False)
        gx = _segment_sum_raw(g1, gi.rowptr_t, None, n)
        _segment_sum_raw(g0, gi.rowptr_s, gi.spos, n, out=gx, accumulate=True)
        return gx, None


def edge_features(x: Tensor, edge_index: Tensor) -> Tensor:
    """``cat[x[e0] - x[e1], x[e0] + x[e1]]`` -> ``[M, 2F]`` (graph_construction.py:386-393);
    differentiable w.r.t. ``x`` (the reference back-propagates through it into the embedding
    network when ``use_embedding_features`` is set and the network is not frozen)."""
    _capi.require_device(x, edge_index)
    if x.dtype != torch.float32:
        x = x.float()
    return _EdgeFeatures.apply(x, edge_index)


# -------------------------------------------------------------------------- BCE
class _BCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, y, src_nodes, pt, pt_thld: float):
        _capi.require_device(w, y)
        lib = _capi.load()
        w = w.contiguous().view(-1)
        y = y.contiguous().view(-1)
        n = w.numel()
        loss = torch.empty(1, dtype=torch.float32, device=w.device)
        ws = _ws(lib.gnntrk_bce_workspace_bytes(n), w)
        _capi.check(lib.gnntrk_bce_forward(_p(w), _p(y), _p(src_nodes), _p(pt), pt_thld, n,
                                           _p(loss), _p(ws), ws.numel(), _stream(w)), lib)
        ctx.save_for_backward(w, y)
        ctx.aux = (src_nodes, pt, pt_thld)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _capi.load()
        w, y = ctx.saved_tensors
        src_nodes, pt, thld = ctx.aux
        g = g.contiguous().view(1).to(torch.float32)
        gw = torch.empty_like(w)
        _capi.check(lib.gnntrk_bce_backward(_p(w), _p(y), _p(src_nodes), _p(pt), thld, w.numel(),
                                            _p(g), _p(gw), _stream(w)), lib)
        return gw, None, None, None, None


class _BCECsr(torch.autograd.Function):
    """BCE of CSR-ordered weights against carried 1-byte CSR labels: loss and the unit gradient in
    one pass (``gnntrk_bce_csr``); the backward is one scaling of the saved unit gradient."""

    @staticmethod
    def forward(ctx, w, label_csr, src_csr, pt, pt_thld: float):
        _capi.require_device(w, label_csr)
        lib = _capi.load()
        w = w.contiguous().view(-1)
        n = w.numel()
        loss = torch.empty(1, dtype=torch.float32, device=w.device)
        gw = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        ws = _ws(lib.gnntrk_bce_workspace_bytes(n), w)
        _capi.check(lib.gnntrk_bce_csr(_p(w), _p(label_csr), _p(src_csr), _p(pt), pt_thld, n, _p(loss), _p(gw),
                                       _p(ws), ws.numel(), _stream(w)), lib)
        # (saved, not kept as an attribute: the E-sized buffer is released with the graph)
        ctx.save_for_backward(*([gw] if gw is not None else []))
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (gw,) = ctx.saved_tensors
        return gw * g.to(torch.float32), None, None, None, None


class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, y, src_nodes, pt, pt_thld: float, alpha: float, gamma: float, pos_weight: float,
                haughty: bool):
        _capi.require_device(w, y)
        lib = _capi.load()
        w = w.contiguous().view(-1)
        y = y.contiguous().view(-1)
        n = w.numel()
        loss = torch.empty(1, dtype=torch.float32, device=w.device)
        ws = _ws(lib.gnntrk_bce_workspace_bytes(n), w)
        _capi.check(lib.gnntrk_focal_forward(_p(w), _p(y), _p(src_nodes), _p(pt), pt_thld, alpha, gamma, pos_weight,
                                             int(haughty), n, _p(loss), _p(ws), ws.numel(), _stream(w)), lib)
        ctx.save_for_backward(w, y)
        ctx.aux = (src_nodes, pt, pt_thld, alpha, gamma, pos_weight, int(haughty))
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _capi.load()
        w, y = ctx.saved_tensors
        src_nodes, pt, thld, alpha, gamma, pos_weight, haughty = ctx.aux
        g = g.contiguous().view(1).to(torch.float32)
        gw = torch.empty_like(w)
        _capi.check(lib.gnntrk_focal_backward(_p(w), _p(y), _p(src_nodes), _p(pt), thld, alpha, gamma, pos_weight,
                                              haughty, w.numel(), _p(g), _p(gw), _stream(w)), lib)
        return gw, None, None, None, None, None, None, None, None


def edge_targets_csr(y: Tensor, gi: GraphIndex, pt: Optional[Tensor] = None, pt_thld: float = 0.0) -> Tensor:
    """The edge labels in ``gi``'s CSR order, pt-falsified (``falsify_low_pt_edges``,
    metrics/losses/ec.py:71-92): one gather per batch, remembered on the graph index (the
    forward and the backward of the loss, and several losses on one batch, share it)."""
    key = (id(y), y._version, None if pt is None else id(pt), float(pt_thld))
    hit = getattr(gi, "_targets", None)
    if hit is not None and hit[0] == key and hit[1]() is y:
        return hit[2]
    _capi.require_device(y)
    lib = _capi.load()
    # 1-byte labels (the dataset's bool y) are gathered as they are; anything else as float
    u8 = y.dtype in (torch.bool, torch.uint8)
    yf = (y.detach().view(torch.uint8) if u8 else y.detach().to(torch.float32)).contiguous().view(-1)
    if yf.numel() != gi.n_edges:
        raise ValueError(f"labels have {yf.numel()} entries, the graph has {gi.n_edges} edges")
    ptf = None
    if pt_thld > 0.0:
        assert pt is not None
        ptf = gi.node_values(pt.detach().to(torch.float32)).contiguous()
    out = torch.empty(gi.n_edges, dtype=torch.float32, device=yf.device)
    _capi.check(lib.gnntrk_edge_targets_csr(_p(yf), int(u8), _p(gi.perm), _p(gi.src), _p(ptf), float(pt_thld),
                                            gi.n_edges, _p(out), _stream(yf)), lib)
    gi._targets = (key, weakref.ref(y), out)
    return out


def _csr_fast_path(w, edge_index):
    """(values in CSR order, graph index) when ``w`` is a model output still held in CSR
    order (edge_order.EdgeOrdered) for the graph of ``edge_index``; else None.

    ``edge_index`` has to be the very tensor the graph index was built from (same object, same
    version): a caller who permutes ``edge_index`` and the labels together gets the ordinary
    path - the labels are gathered through W's own permutation here."""
    from .edge_order import EdgeOrdered

    if not isinstance(w, EdgeOrdered):
        return None
    gi = w.graph_index
    if edge_index is not None:
        if int(edge_index.shape[1]) != gi.n_edges:
            return None
        src = getattr(gi, "_built_from", None)
        if src is not None and not (src[0]() is edge_index and src[1] == edge_index._version):
            return None
    return w.csr.reshape(-1), gi


def focal_loss(w: Tensor, y: Tensor, edge_index: Optional[Tensor] = None, pt: Optional[Tensor] = None,
               pt_thld: float = 0.0, *, alpha: float = 0.25, gamma: float = 2.0, pos_weight: float = 1.0,
               haughty: bool = False) -> Tensor:
    """Binary focal loss of the edge weights (metrics/losses/ec.py:13-68) with the label
    falsification of ``EdgeWeightFocalLoss`` (haughty=False) or ``HaughtyFocalLoss`` (True)."""
    if w.numel() == 0:
        return w.sum() * float("nan")  # the reference's mean over no edges
    if w.dtype != torch.float32:
        raise TypeError("focal_loss: w must be fp32")
    assert gamma >= 0.0
    assert 0 <= alpha <= 1
    fast = None if haughty else _csr_fast_path(w, edge_index)  # (haughty needs raw AND falsified labels)
    if fast is not None:
        w_csr, gi = fast
        t = edge_targets_csr(y, gi, pt, float(pt_thld))
        return _Focal.apply(w_csr, t, None, None, 0.0, float(alpha), float(gamma), float(pos_weight), False)
    y = y.to(torch.float32)
    src_nodes = None
    if pt_thld > 0.0:
        assert edge_index is not None and pt is not None
        src_nodes = edge_index[0].contiguous()
        pt = pt.to(torch.float32).contiguous()
    else:
        pt = None
    return _Focal.apply(w, y, src_nodes, pt, float(pt_thld), float(alpha), float(gamma), float(pos_weight),
                        bool(haughty))


def bce_loss(w: Tensor, y: Tensor, edge_index: Optional[Tensor] = None,
             pt: Optional[Tensor] = None, pt_thld: float = 0.0) -> Tensor:
    """mean BCE(w, y') with y' = falsify_low_pt_edges(y) (metrics/losses/ec.py:71-121)."""
    if w.numel() == 0:
        return w.sum() * float("nan")  # the reference's mean over no edges
    if w.dtype != torch.float32:
        raise TypeError("bce_loss: w must be fp32")
    fast = _csr_fast_path(w, edge_index)
    if fast is not None:  # the weights are still in CSR order: labels go there, W stays put
        w_csr, gi = fast
        lab = carried_label(gi, y)
        if lab is not None:  # ... and the labels came along with the graph-index build: one fused pass
            ptf = None
            if pt_thld > 0.0:
                assert pt is not None
                ptf = pt.detach().to(torch.float32).contiguous()
                if ptf.device != w_csr.device or ptf.numel() < gi.n_nodes:
                    raise ValueError(f"bce_loss: pt must hold one value per node ({gi.n_nodes}) on {w_csr.device}, "
                                     f"got {ptf.numel()} on {ptf.device}")
                ptf = gi.node_values(ptf).contiguous()
            return _BCECsr.apply(w_csr, lab, gi.src if pt_thld > 0.0 else None, ptf, float(pt_thld))
        return _BCE.apply(w_csr, edge_targets_csr(y, gi, pt, float(pt_thld)), None, None, 0.0)
    y = y.to(torch.float32)
    src_nodes = None
    if pt_thld > 0.0:
        assert edge_index is not None and pt is not None
        src_nodes = edge_index[0].contiguous()
        pt = pt.to(torch.float32).contiguous()
    else:
        pt = None
    return _BCE.apply(w, y, src_nodes, pt, float(pt_thld))
