"""bf16-storage variants of the fused ops (BASELINE configs 3/4).

Activations live in HBM as ``torch.bfloat16`` rows padded to a multiple of four
elements (8-byte chunks: one load per lane and chunk in the kernels); parameters,
parameter gradients and the edge weights ``W`` stay fp32; all accumulation is fp32
(include/gnntrk.h: gnntrk_mlp_forward_bf16 / gnntrk_mlp_backward_bf16).

A logical ``[R, dim]`` bf16 tensor is a view ``buf[:, :dim]`` of a ``[R, pad4(dim)]``
buffer, so ``tensor.stride(0)`` carries the padded row stride through autograd.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
from torch import Tensor

from . import _capi

BF16 = torch.bfloat16


_DEBUG_FLAGS = int(__import__("os").environ.get("GNNTRK_DEBUG_FLAGS", "0"))


def pad4(d: int) -> int:
    return (int(d) + 3) // 4 * 4


def empty_rows(n: int, dim: int, device, zero: bool = False) -> Tensor:
    """[n, dim] bf16 view of a padded, 8-byte aligned buffer."""
    mk = torch.zeros if zero else torch.empty
    return mk(int(n), pad4(dim), dtype=BF16, device=device)[:, :dim]


def rows16(t: Tensor) -> Tensor:
    """Make ``t`` a kernel-ready bf16 row tensor (no copy when it already is)."""
    if t.dim() == 1:
        t = t.unsqueeze(1)
    if t.dim() != 2:
        raise ValueError(f"expected a 1-D or 2-D tensor, got shape {tuple(t.shape)}")
    if t.dtype != BF16:
        raise TypeError(f"bf16 path got {t.dtype}")
    ok = ((t.shape[1] == 1 or t.stride(1) == 1) and t.stride(0) % 4 == 0
          and t.stride(0) >= pad4(t.shape[1]) and t.data_ptr() % 8 == 0)
    if t.shape[0] <= 1:
        ok = ok and t.data_ptr() % 8 == 0 and t.stride(0) >= pad4(t.shape[1]) and t.stride(0) % 4 == 0
    if ok:
        return t
    out = empty_rows(t.shape[0], t.shape[1], t.device, zero=True)
    out.copy_(t)
    return out


def to_rows16(x: Tensor, idx: Optional[Tensor] = None) -> Tensor:
    """fp32 rows -> padded bf16 rows (RNE), optionally gathered: ``out[m] = x[idx[m]]``.
    One pass (gnntrk_rows_to_bf16): the dataset's fp32 ``x`` / ``edge_attr`` enter the bf16
    stack through it, ``edge_attr`` already permuted into CSR order."""
    from . import ops
    lib = _capi.load()
    x = ops._as_rows(x.detach())
    m = int(idx.shape[0]) if idx is not None else int(x.shape[0])
    out = empty_rows(m, x.shape[1], x.device)
    _capi.check(lib.gnntrk_rows_to_bf16(ops._p(x), x.shape[1], ops._row_stride(x), ops._p(idx), m,
                                        ops._p(out), out.stride(0), ops._stream(x)), lib)
    return out


def _seg16(t: Tensor, idx, relu: bool):
    return _capi.Seg(t.data_ptr(), None if idx is None else idx.data_ptr(), t.shape[1], t.stride(0),
                     int(relu), min(int(t.shape[0]), 0x7fffffff))


def mlp_forward_raw(segs: Sequence[Tensor], idx: Sequence[Optional[Tensor]], relu: Sequence[bool],
                    weights: Sequence[Tensor], biases: Sequence[Optional[Tensor]], *, n_rows: int,
                    epilogue: int, ca: float, cb: float, res: Optional[Tensor],
                    out_idx: Optional[Tensor], out_rows: int, mlp) -> Tensor:
    """One launch of gnntrk_mlp_forward_bf16; ``segs``/``res`` kernel-ready (rows16)."""
    from . import ops
    lib = _capi.load()
    a = _capi.MlpFwdArgs()
    a.mlp = mlp
    a.n_seg, a.epilogue, a.n_rows = len(segs), epilogue, n_rows
    for j, s in enumerate(segs):
        a.seg[j] = _seg16(s, idx[j], relu[j])
    a.ca, a.cb = ca, cb
    if epilogue == _capi.EPI_RESIDUAL:
        a.res, a.res_stride = res.data_ptr(), res.stride(0)
    dev = segs[0].device
    if epilogue == _capi.EPI_SIGMOID:
        out = torch.empty(out_rows, mlp.out_dim, dtype=torch.float32, device=dev)
    else:
        out = empty_rows(out_rows, mlp.out_dim, dev)
    a.out, a.out_stride, a.out_idx = out.data_ptr(), out.stride(0), ops._p(out_idx)
    a.debug_flags = _DEBUG_FLAGS & 4096   # (4096: the I/O skeleton of the two large forward shapes - bench.py's access floor)
    M = n_rows
    nbytes = M * (sum(2 * s.shape[1] + (4 if idx[j] is not None else 0) for j, s in enumerate(segs))
                  + (4 if epilogue == _capi.EPI_SIGMOID else 2) * mlp.out_dim
                  + (4 if out_idx is not None else 0)
                  + (2 * mlp.out_dim if epilogue == _capi.EPI_RESIDUAL else 0))
    key = ""
    if ops._TIMER is not None:
        buf = C.create_string_buffer(160)
        _capi.check(lib.gnntrk_mlp_forward_bf16_kernel_name(C.byref(a), buf, len(buf)), lib)
        key = buf.value.decode()
    with ops._timed(out, key, ops._mlp_flops_per_row(mlp) * M, nbytes, M):
        _capi.check(lib.gnntrk_mlp_forward_bf16(C.byref(a), ops._stream(out)), lib)
    return out


#: on: the gradient of a target-gathered segment leaves the backward kernel summed per node (include/gnntrk.h:
#: gnntrk_gfold) where the launch takes it.  OFF by default: built in round 6, parity green (case_mlp_bf16_fold),
#: and measured slower - the segment matrix, five more MFMAs and their packs cost the relational backward
#: +0.35 ms per 64 M rows (2.34 -> 2.68-2.76) and the head +0.24 ms (3.02 -> 3.26), more than the 0.23 ms streaming
#: fold + the 16 B/row store they replace: cfg3 step 24.1 against 23.4 ms
FOLD_IN_KERNEL = __import__("os").environ.get("GNNTRK_FOLD_IN_KERNEL", "0") != "0"


def mlp_backward_raw(segs: Sequence[Tensor], idx: Sequence[Optional[Tensor]], relu: Sequence[bool],
                     weights: Sequence[Tensor], biases: Sequence[Optional[Tensor]], *, n_rows: int,
                     epilogue: int, ca: float, cb: float, gout: Sequence[tuple], need_seg: Sequence[bool],
                     want_dw: bool, mlp, gidx: Optional[Sequence[Optional[Tensor]]] = None, sinks=None,
                     fold: Optional[tuple] = None):
    """One launch of gnntrk_mlp_backward_bf16.  ``gout``: 1-2 tuples (rows, idx) - padded
    bf16 rows, or one fp32 ``[*, out]`` tensor for EPI_SIGMOID.  Returns (row-aligned
    per-segment gradient slices ``[n_rows, dim]`` bf16 or None, gW list, gb list).
    ``fold = (j, n_nodes, rowptr)``: segment ``j`` is gathered through sorted ids (the CSR targets; ``rowptr``: their
    int32 row pointers ``[n_nodes + 1]``) - where the launch takes it (``gnntrk_mlp_backward_bf16_can_fold``) its
    gradient comes back ALREADY SUMMED per node as ``[n_nodes, dim]`` rows in ``slices[j]`` and ``j`` is listed in
    the returned ``slices.folded`` set."""
    from . import ops
    lib = _capi.load()
    a = _capi.MlpBwdArgs()
    a.mlp = mlp
    a.n_seg, a.epilogue, a.n_rows = len(segs), epilogue, n_rows
    a.ca, a.cb = ca, cb
    for j, s in enumerate(segs):
        a.seg[j] = _seg16(s, idx[j], relu[j])
    dev = segs[0].device
    slices = _Slices([None] * len(segs))
    for j, s in enumerate(segs):
        if need_seg[j]:
            slices[j] = empty_rows(n_rows, s.shape[1], dev)
            gi_j = None if gidx is None else gidx[j]
            a.gseg[j] = _capi.GSeg(slices[j].data_ptr(), ops._p(gi_j), slices[j].stride(0), 0)
    fold_bufs = None

    def try_fold():
        """Arm the in-kernel fold of segment ``fold[0]`` if this launch takes it (asked once the terms are set)."""
        nonlocal fold_bufs
        if fold is None or not FOLD_IN_KERNEL or _DEBUG_FLAGS:
            return
        j, n_nodes, rowptr = fold
        if not need_seg[j] or idx[j] is None or segs[j].shape[1] > 8 or n_rows < 1 or n_nodes < 1:
            return
        if segs[j].stride(0) != 8 or segs[j].data_ptr() % 16 or any(bool(r) != bool(relu[j]) for k, r in enumerate(relu) if need_seg[k]):
            return   # (the finishing pass gates with 16-byte rows of the segment itself; one ReLU flag for all slices)
        if gidx is not None and gidx[j] is not None:
            return
        units = (n_rows + 31) // 32
        # (node rows, zero-filled, and the units' carry rows behind them: one allocation, one descriptor in the kernel)
        out = empty_rows(n_nodes + units, segs[j].shape[1], dev, zero=True)
        if out.stride(0) != 8:
            return
        old = (a.gseg[j].ptr, a.gseg[j].idx, a.gseg[j].stride, a.gseg[j].accumulate)
        a.gseg[j] = _capi.GSeg(out.data_ptr(), None, 8, 0)
        a.fold = _capi.GFold(idx[j].data_ptr(), n_nodes, j, 0)
        if int(lib.gnntrk_mlp_backward_bf16_can_fold(C.byref(a))) != 1:
            a.gseg[j] = _capi.GSeg(*old)
            a.fold = _capi.GFold(None, 0, -1, 0)
            return
        fold_bufs = (j, out, rowptr, units, n_nodes)

    def set_terms(terms):
        a.n_gout = len(terms)
        for t, (rows, term_idx) in enumerate(terms):
            a.gout[t] = _capi.GTerm(rows.data_ptr(), ops._p(term_idx), rows.stride(0), min(int(rows.shape[0]), 0x7fffffff))

    gout = list(gout)
    set_terms(gout[:2])
    a.debug_flags = _DEBUG_FLAGS
    # more terms than this launch's kernel takes (three only on the buffer-addressed shapes): the
    # surplus rows-of-the-tile terms are summed first, as autograd would have done
    limit = 2 if epilogue == _capi.EPI_SIGMOID else (int(lib.gnntrk_mlp_backward_bf16_max_terms(C.byref(a))) if len(gout) > 2 else 2)
    while len(gout) > limit:
        (r1, i1), (r2, i2) = gout[-2], gout[-1]
        j = len(gout) - 2
        if i1 is not None or i2 is not None:   # fold two plain terms; a gathered one stays as it is
            plain = [k for k, (_, ix) in enumerate(gout) if ix is None]
            if len(plain) < 2:
                raise RuntimeError("mlp_backward_raw: cannot fold gathered upstream terms")
            j, k2 = plain[-2], plain[-1]
            (r1, i1), (r2, i2) = gout[j], gout[k2]
            del gout[k2]
        else:
            del gout[-1]
        gout[j] = (rows16(r1 + r2), None)
    set_terms(gout)
    try_fold()
    gW = [None] * len(weights)
    gb = [None] * len(weights)
    if want_dw:
        if sinks is not None:   # (persistent fp32 gradient buffers of the parameters: ops._param_grad_sinks)
            gW, gb = sinks
        else:
            gW = [torch.empty_like(w) for w in weights]
            gb = [None if b is None else torch.empty_like(b) for b in biases]
        for i in range(len(weights)):
            a.gW[i] = gW[i].data_ptr()
            a.gb[i] = ops._p(gb[i])
    # always needed: per-wave partial blocks + the store-redirect slots of masked lanes
    ws = ops._ws(lib.gnntrk_mlp_backward_bf16_workspace_bytes(C.byref(a.mlp)), segs[0])
    a.accumulate_params = 1 if (want_dw and sinks is not None) else 0
    # (debug_flags: 64 one 16-row tile per iteration instead of two, 128 generic per-lane I/O - A/B timing)
    M = n_rows
    # (algorithmic bytes as SURVEY 8d counts them: a gathered row and its gradient per edge, whether or not the
    #  gradient leaves this kernel already folded)
    nbytes = M * (sum(2 * s.shape[1] + (4 if idx[j] is not None else 0)
                      + (2 * s.shape[1] if need_seg[j] else 0) for j, s in enumerate(segs))
                  + sum((4 if epilogue == _capi.EPI_SIGMOID else 2) * mlp.out_dim
                        + (4 if gi is not None else 0) for _, gi in gout))
    ref = gout[0][0]
    key = ""
    if ops._TIMER is not None:
        buf = C.create_string_buffer(160)
        _capi.check(lib.gnntrk_mlp_backward_bf16_kernel_name(C.byref(a), buf, len(buf)), lib)
        key = buf.value.decode()
    with ops._timed(ref, key, 3 * ops._mlp_flops_per_row(mlp) * M, nbytes, M):
        _capi.check(lib.gnntrk_mlp_backward_bf16(C.byref(a), ops._p(ws), 0 if ws is None else ws.numel(),
                                                 ops._stream(ref)), lib)
    if fold_bufs is not None:
        j, out, rowptr, units, n_nodes = fold_bufs
        gate = segs[j] if relu[j] else None   # (relu' of the segment: once per node, by its own input row)
        _capi.check(lib.gnntrk_fold_finish_bf16(out.data_ptr(), 8, n_nodes, ops._p(rowptr), units,
                                                None if gate is None else gate.data_ptr(), 8, ops._stream(ref)), lib)
        slices[j] = out[:n_nodes]
        slices.folded.add(j)
    return slices, gW, gb


class _Slices(list):
    """The per-segment gradient slices of one backward launch; ``folded``: segments whose entry is already the
    per-node sum (in-kernel fold) instead of one row per edge."""

    def __init__(self, it):
        super().__init__(it)
        self.folded = set()


# ------------------------------------------------------------------ plain helpers
def segment_sum_raw(rows: Tensor, rowptr: Tensor, pos: Optional[Tensor], n_seg: int,
                    addend: Optional[Tensor] = None) -> Tensor:
    """``out[n] = bf16(sum of rows over segment n [+ addend[n]])`` - fp32 accumulation, one rounding."""
    from . import ops
    lib = _capi.load()
    rows = rows16(rows)
    out = empty_rows(n_seg, rows.shape[1], rows.device)
    if addend is None:
        _capi.check(lib.gnntrk_segment_sum_bf16(rows.data_ptr(), rows.shape[1], rows.stride(0), ops._p(rowptr),
                                                ops._p(pos), n_seg, out.data_ptr(), out.stride(0),
                                                ops._stream(rows)), lib)
        return out
    addend = rows16(addend)
    if addend.shape != (n_seg, rows.shape[1]):
        raise ValueError(f"segment_sum addend must be [{n_seg}, {rows.shape[1]}], got {tuple(addend.shape)}")
    _capi.check(lib.gnntrk_segment_sum_bf16_add(rows.data_ptr(), rows.shape[1], rows.stride(0), ops._p(rowptr),
                                                ops._p(pos), n_seg, addend.data_ptr(), addend.stride(0),
                                                out.data_ptr(), out.stride(0), ops._stream(rows)), lib)
    return out


def permute_raw(x: Tensor, idx: Tensor, scatter: bool) -> Tensor:
    from . import ops
    lib = _capi.load()
    x2 = rows16(x)
    m = int(idx.shape[0])
    if scatter and m != x2.shape[0]:
        raise ValueError("permute_rows(scatter): idx must be a permutation of the rows")
    out = empty_rows(m, x2.shape[1], x2.device)
    _capi.check(lib.gnntrk_permute_rows_bf16(x2.data_ptr(), x2.shape[1], x2.stride(0), ops._p(idx), m,
                                             out.data_ptr(), out.stride(0), int(scatter),
                                             ops._stream(x2)), lib)
    return out.view(-1) if x.dim() == 1 else out


class SegmentSum16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, gi, by: str):
        _capi.require_device(rows)
        ctx.gi, ctx.by = gi, by
        if by == "tgt":
            return segment_sum_raw(rows, gi.rowptr_t, None, gi.n_nodes)
        if by == "src":
            return segment_sum_raw(rows, gi.rowptr_s, gi.spos, gi.n_nodes)
        raise ValueError(by)

    @staticmethod
    def backward(ctx, g):
        gi = ctx.gi
        return permute_raw(g, gi.tgt if ctx.by == "tgt" else gi.src, scatter=False), None, None


class PermuteRows16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, scatter: bool):
        _capi.require_device(x)
        ctx.idx, ctx.scatter = idx, scatter
        return permute_raw(x, idx, scatter)

    @staticmethod
    def backward(ctx, g):
        return permute_raw(g, ctx.idx, not ctx.scatter), None, None


# ------------------------------------------------------------------ fused MLP (autograd)
class FusedMLP16(torch.autograd.Function):
    """bf16-storage twin of ops._FusedMLP (same spec object, same reduce rules)."""

    @staticmethod
    def forward(ctx, spec, *tensors):
        from . import ops
        ns, nl = spec.n_seg, spec.n_layers
        segs = [rows16(t) for t in tensors[:ns]]
        weights = [w.contiguous() for w in tensors[ns:ns + nl]]
        biases = [None if b is None else b.contiguous() for b in tensors[ns + nl:ns + 2 * nl]]
        res = tensors[ns + 2 * nl]
        _capi.require_device(*segs, *weights)
        mlp = ops._fill_mlp(weights, biases)
        if sum(s.shape[1] for s in segs) != mlp.in_dim:
            raise AssertionError(
                f"Expected feature dimension {mlp.in_dim}, got {sum(s.shape[1] for s in segs)}")
        ctx.res_key = None
        ctx.dup_of, ctx.res_same = _alias_tables(tensors[:ns], res if spec.epilogue == _capi.EPI_RESIDUAL else None)
        if spec.epilogue == _capi.EPI_RESIDUAL:
            res = rows16(res)
            ctx.res_key = (res.data_ptr(), tuple(res.shape), res.stride(0))
        out = mlp_forward_raw(segs, spec.idx, spec.relu, weights, biases, n_rows=spec.n_rows,
                              epilogue=spec.epilogue, ca=spec.ca, cb=spec.cb, res=res,
                              out_idx=spec.out_idx, out_rows=spec.out_rows, mlp=mlp)
        ctx.spec = spec
        ctx.save_for_backward(*segs, *weights, *[b for b in biases if b is not None])
        ctx.bias_mask = [b is not None for b in biases]
        ctx.stash = None
        if spec.epilogue != _capi.EPI_SIGMOID and spec.out_idx is None and spec.epilogue != _capi.EPI_RESIDUAL:
            _attach_stash(ctx, out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        from . import ops
        spec = ctx.spec
        extra = _stashed_terms(ctx)
        if g_out is None:   # (every reader went through a tap)
            if not extra:
                raise RuntimeError("FusedMLP16.backward without upstream gradients")
            g_rows = extra.pop(0)[0]
        elif spec.epilogue == _capi.EPI_SIGMOID:
            g_rows = ops._as_rows(g_out.float().contiguous())
        else:
            g_rows = rows16(g_out)
        outs, g_res = _backward_common(ctx, [(g_rows, spec.out_idx)] + extra, ctx.needs_input_grad, g_rows)
        return (None, *outs, g_res)


def _tensor_key(t: Tensor):
    return (t.data_ptr(), tuple(t.shape), t.stride(0))


def _same_autograd(a, b) -> bool:
    """The two inputs are ONE tensor for autograd's purposes: the same object, or one is the identity view this
    module itself made of the other (``node_tap`` / ``grad_tap`` mark their outputs).  Two unrelated autograd tensors
    that merely alias memory (``x`` and ``x.detach().requires_grad_()``) are not: their gradients stay apart."""
    if a is None or b is None:
        return False
    return a is b or getattr(a, "_gnntrk_tap_of", None) is b or getattr(b, "_gnntrk_tap_of", None) is a


def _alias_tables(seg_tensors, res=None):
    """(dup_of, res_same): for every segment the first earlier segment that is the same autograd tensor (-1: none),
    and the segments that are the same autograd tensor as the residue - recorded at forward time, where the
    tensors' identity is still known (the saved tensors of the backward are unpacked copies)."""
    ts = list(seg_tensors)
    dup_of = [next((k for k in range(j) if _same_autograd(ts[j], ts[k])), -1) for j in range(len(ts))]
    return dup_of, [j for j in range(len(ts)) if _same_autograd(res, ts[j])]


def _backward_common(ctx, gout, need, g_rows, node_addend=None):
    """Shared by FusedMLP16 and FusedINEdge16: one gnntrk_mlp_backward_bf16 launch for the
    upstream terms ``gout`` + the folds of the gathered input gradients.  ``need`` is laid
    out as [spec, segs..., W..., b..., res].  Returns (grads of segs/W/b, grad of res)."""
    from . import ops
    spec = ctx.spec
    ns, nl = spec.n_seg, spec.n_layers
    saved = ctx.saved_tensors
    segs, weights = list(saved[:ns]), list(saved[ns:ns + nl])
    bl = list(saved[ns + nl:ns + nl + sum(ctx.bias_mask)])
    biases = [bl.pop(0) if m else None for m in ctx.bias_mask]
    mlp = ops._fill_mlp(weights, biases)
    M = spec.n_rows
    need_seg = [bool(need[1 + j]) for j in range(ns)]
    for j in range(ns):
        if need_seg[j] and spec.idx[j] is not None and spec.reduce[j] is None:
            raise RuntimeError("gathered segment requires a `reduce` rule for backward")
    want_dw = any(need[1 + ns:1 + ns + 2 * nl])
    # source-gathered node rows: the kernel writes their per-edge gradients already in
    # source-sorted order (a permuted 16-byte store), so the fold below streams instead of
    # gathering random rows
    gidx = [spec.reduce[j][1].spos_inv
            if (need_seg[j] and isinstance(spec.reduce[j], tuple) and spec.reduce[j][0] == "src")
            else None for j in range(ns)]
    sinks = None
    if want_dw:
        sinks = ops._param_grad_sinks(weights, biases, need[1 + ns:1 + ns + nl],
                                      [need[1 + ns + nl + i] or biases[i] is None for i in range(nl)])
    # the target-gathered segment (CSR order = sorted by target): folded inside the kernel where the launch takes it
    fold = next(((j, int(segs[j].shape[0]), spec.reduce[j][1].rowptr_t) for j in range(ns)
                 if need_seg[j] and isinstance(spec.reduce[j], tuple) and spec.reduce[j][0] == "tgt"
                 and spec.idx[j] is spec.reduce[j][1].tgt), None)
    slices, gW, gb = mlp_backward_raw(segs, spec.idx, spec.relu, weights, biases, n_rows=M,
                                      epilogue=spec.epilogue, ca=spec.ca, cb=spec.cb, gout=gout,
                                      need_seg=need_seg, want_dw=want_dw, mlp=mlp, gidx=gidx, sinks=sinks, fold=fold)
    seg_grads = [None] * ns
    folded: dict = {}   # (tensor identity) -> index of the segment whose fold holds its gradient so far
    # (segments the kernel folded itself come first: a later fold of the same tensor then takes the result as its
    #  fp32 addend - one pass, one rounding - instead of a torch add of two N-sized tensors)
    order = sorted(range(ns), key=lambda j: (j not in slices.folded, j))
    for j in order:
        s = segs[j]
        if slices[j] is None:
            continue
        if spec.idx[j] is None:
            if s.shape[0] != M:
                raise RuntimeError("identity segments must have n_rows rows")
            seg_grads[j] = slices[j]
        elif spec.reduce[j] == "perm":
            if s.shape[0] != M:
                raise RuntimeError("'perm' segments must cover all source rows")
            seg_grads[j] = permute_raw(slices[j], spec.idx[j], scatter=True)
        else:
            by, gi = spec.reduce[j]
            rowptr = gi.rowptr_t if by == "tgt" else gi.rowptr_s  # (src rows are pre-sorted)
            # a tensor gathered twice (the node embedding by target AND by source): the second fold takes the
            # first as one more fp32 term, and the sum leaves as ONE gradient with one rounding - autograd would
            # add the two (an N-sized pass, a second rounding, and a re-padding copy of its dense result)
            # (same memory AND the same autograd tensor: two unrelated tensors that alias keep separate gradients)
            first = folded.get(_tensor_key(s)) if FOLD_ADD else None
            if (first is not None and getattr(ctx, "dup_of", None) is not None and ctx.dup_of[j] != first
                    and ctx.dup_of[first] != j):   # (either order: the kernel-folded segment goes first)
                first = None
            if j in slices.folded:
                # summed per node by the backward kernel itself (gnntrk_gfold): nothing to read back per edge
                seg_grads[j] = slices[j]
                if first is not None:
                    seg_grads[j] = rows16(seg_grads[j] + seg_grads[first])
                    seg_grads[first] = None
                elif node_addend is not None and node_addend[0] == _tensor_key(s):
                    seg_grads[j] = rows16(seg_grads[j] + rows16(node_addend[1]))
                    node_addend = None
            elif first is not None:
                seg_grads[j] = segment_sum_raw(slices[j], rowptr, None, s.shape[0], addend=seg_grads[first])
                seg_grads[first] = None
            elif node_addend is not None and node_addend[0] == _tensor_key(s):
                # (the object model's gradient of the same embedding, handed over by node_tap)
                seg_grads[j] = segment_sum_raw(slices[j], rowptr, None, s.shape[0], addend=node_addend[1])
                node_addend = None
            else:
                seg_grads[j] = segment_sum_raw(slices[j], rowptr, None, s.shape[0])
            folded[_tensor_key(s)] = j
    g_res = None
    if spec.epilogue == _capi.EPI_RESIDUAL and need[1 + ns + 2 * nl]:
        g_dense = g_rows if spec.out_idx is None else permute_raw(g_rows, spec.out_idx, False)
        res_key = getattr(ctx, "res_key", None)
        same = [j for j, s in enumerate(segs) if FOLD_ADD and res_key is not None and spec.idx[j] is None
                and seg_grads[j] is not None and (s.data_ptr(), tuple(s.shape), s.stride(0)) == res_key
                and seg_grads[j].shape == g_dense.shape and j in getattr(ctx, "res_same", (j,))]
        if same:
            # the residue IS an identity segment of this node (resin.py:26: the layer's own input): its
            # pass-through gradient joins that segment's gradient in one pass instead of a mul and autograd's add
            seg_grads[same[0]].add_(g_dense, alpha=spec.ca)
        else:
            g_res = g_dense * spec.ca
    if node_addend is not None:
        raise RuntimeError("a node gradient handed over by node_tap found no fold to join")
    outs = list(seg_grads)
    if sinks is not None:   # (already added into the parameters' gradient buffers)
        outs += [None] * (2 * nl)
    else:
        outs += [gW[i] if need[1 + ns + i] else None for i in range(nl)]
        outs += [gb[i] if need[1 + ns + nl + i] else None for i in range(nl)]
    return outs, g_res


class FusedINEdge16(torch.autograd.Function):
    """Relational model + sum aggregation of one interaction-network layer
    (interaction_network.py:67-89) as ONE autograd node: ``(e~, aggr) = f(segments, params)``.
    The backward feeds the kernel both upstream terms - ``g_e~[k] + g_aggr[tgt[k]]`` - so the
    gathered aggregation gradient and the sum are never materialised (include/gnntrk.h:
    gout[2])."""

    @staticmethod
    def forward(ctx, spec, gi, *tensors):
        from . import ops
        ns, nl = spec.n_seg, spec.n_layers
        segs = [rows16(t) for t in tensors[:ns]]
        weights = [w.contiguous() for w in tensors[ns:ns + nl]]
        biases = [None if b is None else b.contiguous() for b in tensors[ns + nl:ns + 2 * nl]]
        _capi.require_device(*segs, *weights)
        mlp = ops._fill_mlp(weights, biases)
        if sum(s.shape[1] for s in segs) != mlp.in_dim:
            raise AssertionError(
                f"Expected feature dimension {mlp.in_dim}, got {sum(s.shape[1] for s in segs)}")
        e_tilde = mlp_forward_raw(segs, spec.idx, spec.relu, weights, biases, n_rows=spec.n_rows,
                                  epilogue=spec.epilogue, ca=spec.ca, cb=spec.cb, res=None,
                                  out_idx=None, out_rows=spec.out_rows, mlp=mlp)
        aggr = segment_sum_raw(e_tilde, gi.rowptr_t, None, gi.n_nodes)
        ctx.spec, ctx.gi = spec, gi
        ctx.dup_of, ctx.res_same = _alias_tables(tensors[:ns])
        ctx.save_for_backward(*segs, *weights, *[b for b in biases if b is not None])
        ctx.bias_mask = [b is not None for b in biases]
        _attach_stash(ctx, e_tilde)
        # the node embedding this layer gathers twice (by target and by source): the object model, which reads
        # the same embedding next to `aggr`, can hand its gradient to THIS node's backward (node_tap below)
        ctx.node_stash, ctx.node_key = None, None
        twice = [j for j in range(ns) if isinstance(spec.reduce[j], tuple) and spec.reduce[j][0] in ("tgt", "src")]
        if len(twice) == 2 and _tensor_key(segs[twice[0]]) == _tensor_key(segs[twice[1]]):
            ctx.node_stash, ctx.node_key = _GradStash(), _tensor_key(segs[twice[0]])
            aggr._gnntrk_node_stash = (ctx.node_stash, ctx.node_key)
        return e_tilde, aggr

    @staticmethod
    def backward(ctx, g_et, g_aggr):
        gi = ctx.gi
        gout = []
        if g_et is not None:
            gout.append((rows16(g_et), None))
        if g_aggr is not None:
            gout.append((rows16(g_aggr), gi.tgt))
        gout += _stashed_terms(ctx)   # (a second reader's gradient of e~, handed over by grad_tap)
        if not gout:
            raise RuntimeError("FusedINEdge16.backward without upstream gradients")
        need = ctx.needs_input_grad  # [spec, gi, segs..., W..., b...]
        node_g = ctx.node_stash.take() if ctx.node_stash is not None else None
        outs, _ = _backward_common(ctx, gout, (need[0],) + tuple(need[2:]) + (False,), None,
                                   node_addend=None if node_g is None else (ctx.node_key, node_g))
        return (None, None, *outs)


# ---- gradient tap: a second reader of an edge embedding hands its gradient to the producer directly --
class _GradStash:
    """Filled by ``_Tap.backward`` (the reader's gradient), emptied by the producer's backward."""

    __slots__ = ("g",)

    def __init__(self):
        self.g = None

    def take(self):
        g, self.g = self.g, None
        return g


class _Tap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, stash):
        ctx.stash = stash
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        ctx.stash.g = g if ctx.stash.g is None else ctx.stash.g + g
        return None, None


#: the folds of a tensor gathered twice leave as one sum (off: autograd adds the two)
FOLD_ADD = __import__("os").environ.get("GNNTRK_FOLD_ADD", "1") != "0"

#: gradient taps on / off (off: autograd sums the two gradients of a twice-read embedding itself)
TAP = __import__("os").environ.get("GNNTRK_GRAD_TAP", "1") != "0"


def grad_tap(t: Tensor) -> Tensor:
    """``t`` for a SECOND reader (the edge-weight head reads every edge embedding that the next
    interaction network also reads): the reader's gradient does not go through autograd's sum of
    the two gradients - a 24 B/edge pass per embedding - but into a stash the backward of ``t``'s
    producer (``FusedMLP16`` / ``FusedINEdge16``) picks up as one more upstream term of its kernel.
    Tensors not produced by those nodes, and fp32 ones, are returned as they are."""
    stash = getattr(t, "_gnntrk_stash", None)
    if stash is None or not TAP or not t.requires_grad:
        return t
    return _Tap.apply(t, stash)


class _NodeTap(torch.autograd.Function):
    """Identity on ``(x, aggr)``.  Backward: the gradient of ``x`` goes into the stash, the gradient of ``aggr``
    passes through - so the producer of ``aggr`` can only run its backward AFTER the stash is filled (a true
    dependency, not an ordering habit of the engine)."""

    @staticmethod
    def forward(ctx, x, aggr, stash):
        ctx.stash = stash
        return x.view_as(x), aggr.view_as(aggr)

    @staticmethod
    def backward(ctx, gx, ga):
        if gx is not None:
            ctx.stash.g = gx if ctx.stash.g is None else ctx.stash.g + gx
        return None, ga, None


def node_tap(x: Tensor, aggr: Tensor):
    """``(x, aggr)`` for the object model of an interaction-network layer whose relational model + aggregation
    node (``FusedINEdge16``, the producer of ``aggr``) gathers the same ``x`` by target and by source: the object
    model's gradient of ``x`` goes into that node's backward - which can only run once the gradient of ``aggr`` has
    come through the same tap - and joins its first fold as one more fp32 term, instead of autograd adding two
    gradients of ``x`` (an N-sized pass plus the re-padding copy of its dense result)."""
    held = getattr(aggr, "_gnntrk_node_stash", None)
    if (held is None or not TAP or not FOLD_ADD or not x.requires_grad or not aggr.requires_grad
            or x.dtype != BF16 or x.dim() != 2):
        return x, aggr
    xr = rows16(x)
    if xr is not x or _tensor_key(xr) != held[1]:
        return x, aggr
    xt, at = _NodeTap.apply(x, aggr, held[0])
    xt._gnntrk_tap_of = x   # (an identity view of x made here: the same tensor as far as gradient folds go)
    return xt, at


def _attach_stash(ctx, out: Tensor) -> None:
    ctx.stash = _GradStash()
    out._gnntrk_stash = ctx.stash


def _stashed_terms(ctx):
    g = ctx.stash.take() if getattr(ctx, "stash", None) is not None else None
    return [] if g is None else [(rows16(g), None)]


def in_edge(segs, weights, biases, gi, n_rows: int):
    """``(e~, aggr)`` of one interaction-network layer in bf16 storage (see FusedINEdge16)."""
    from . import ops
    spec = ops._MlpSpec(len(segs), len(weights), any(b is not None for b in biases),
                        [s.idx for s in segs], [s.relu for s in segs], [s.reduce for s in segs],
                        _capi.EPI_NONE, 0.0, 1.0, None, int(n_rows), int(n_rows))
    return FusedINEdge16.apply(spec, gi, *[s.t for s in segs], *weights, *biases)
