"""ctypes binding of the C ABI in ``include/gnntrk.h`` (libgnntrk.so, HIP/gfx950).

This module is the only place that touches the shared library.  There is NO
fallback: if the library cannot be loaded (not built, no ROCm runtime) every op of
the package raises ``RuntimeError`` - the product path never computes on the CPU.
"""

from __future__ import annotations

import ctypes as C
import os
import pathlib
import threading

MAX_SEGS = 10
MAX_IN, MAX_HIDDEN, MAX_OUT = 48, 64, 16            # fp32 kernels
MAX_OUT_BF16 = 48                                  # bf16 storage, three hidden tiles: up to three output tiles
MAX_IN_BF16, MAX_HIDDEN_BF16 = 128, 128             # bf16 storage: 16 four-feature chunks; hidden + bias row <= 96
#                                                    (<= 128 where the inputs fit one k-step of 32 slots)
EPI_NONE, EPI_RELU, EPI_RESIDUAL, EPI_SIGMOID = 0, 1, 2, 3

_PKG = pathlib.Path(__file__).resolve().parent
LIB_PATH = _PKG / "libgnntrk.so"


class Seg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("idx", C.c_void_p), ("dim", C.c_int32),
                ("stride", C.c_int32), ("relu", C.c_int32), ("rows", C.c_int32)]


class Mlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("in_dim", C.c_int32), ("hidden", C.c_int32),
                ("out_dim", C.c_int32), ("W", C.c_void_p * 3), ("b", C.c_void_p * 3)]


class MlpFwdArgs(C.Structure):
    _fields_ = [("mlp", Mlp), ("n_seg", C.c_int32), ("epilogue", C.c_int32),
                ("seg", Seg * MAX_SEGS), ("n_rows", C.c_int64), ("ca", C.c_float),
                ("cb", C.c_float), ("res", C.c_void_p), ("res_stride", C.c_int32),
                ("out_stride", C.c_int32), ("out", C.c_void_p), ("out_idx", C.c_void_p),
                ("debug_flags", C.c_int32), ("_pad", C.c_int32)]


class GTerm(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("idx", C.c_void_p), ("stride", C.c_int32),
                ("rows", C.c_int32)]


class GSeg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("idx", C.c_void_p), ("stride", C.c_int32),
                ("accumulate", C.c_int32)]


class GFold(C.Structure):
    _fields_ = [("ids", C.c_void_p), ("n_nodes", C.c_int64), ("seg", C.c_int32), ("_pad", C.c_int32)]


class MlpBwdArgs(C.Structure):
    _fields_ = [("mlp", Mlp), ("n_seg", C.c_int32), ("epilogue", C.c_int32),
                ("seg", Seg * MAX_SEGS), ("n_rows", C.c_int64), ("ca", C.c_float),
                ("cb", C.c_float), ("n_gout", C.c_int32), ("accumulate_params", C.c_int32),
                ("gout", GTerm * 3), ("gseg", GSeg * MAX_SEGS), ("gW", C.c_void_p * 3),
                ("gb", C.c_void_p * 3), ("debug_flags", C.c_int32), ("_pad", C.c_int32), ("fold", GFold)]


class GraphIndex(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_edges", C.c_int64), ("perm", C.c_void_p),
                ("tgt", C.c_void_p), ("src", C.c_void_p), ("rowptr_t", C.c_void_p),
                ("rowptr_s", C.c_void_p), ("spos", C.c_void_p), ("spos_inv", C.c_void_p)]


class GraphIndexCarry(C.Structure):
    _fields_ = [("edge_label", C.c_void_p), ("label_csr", C.c_void_p), ("edge_rows", C.c_void_p),
                ("rows_csr_bf16", C.c_void_p), ("rows_stride", C.c_int32), ("out_stride", C.c_int32),
                ("node_rank", C.c_void_p)]


class OcArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("beta", C.c_void_p), ("particle_id", C.c_void_p),
                ("mask", C.c_void_p), ("gid", C.c_void_p), ("alphas", C.c_void_p),
                ("n_cp", C.c_void_p), ("n", C.c_int64), ("dim", C.c_int32), ("stride", C.c_int32),
                ("q_min", C.c_float), ("radius", C.c_float), ("eps_sqrt", C.c_float),
                ("mode", C.c_int32), ("rep_keep_prob", C.c_float), ("_pad", C.c_int32),
                ("rep_seed", C.c_uint64), ("cap_nbr", C.c_void_p)]


RESFCNN_MAX_HIDDEN, RESFCNN_MAX_IN, RESFCNN_MAX_WIDTH, RESFCNN_MAX_OUT = 16, 64, 128, 32
WIDE_MAX_IN, WIDE_MAX_HIDDEN, WIDE_MAX_OUT = 128, 128, 48   # gnntrk_mlp_*_wide (fp32)


class ResFcnn(C.Structure):
    _fields_ = [("W_enc", C.c_void_p), ("b_enc", C.c_void_p), ("W_hid", C.c_void_p * RESFCNN_MAX_HIDDEN),
                ("b_hid", C.c_void_p * RESFCNN_MAX_HIDDEN), ("W_dec", C.c_void_p), ("b_dec", C.c_void_p),
                ("out_scale", C.c_void_p), ("in_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32),
                ("n_hidden", C.c_int32), ("alpha", C.c_float), ("normalize", C.c_int32), ("out_relu", C.c_int32),
                ("_pad", C.c_int32)]


class ResFcnnGrads(C.Structure):
    _fields_ = [("W_enc", C.c_void_p), ("b_enc", C.c_void_p), ("W_hid", C.c_void_p * RESFCNN_MAX_HIDDEN),
                ("b_hid", C.c_void_p * RESFCNN_MAX_HIDDEN), ("W_dec", C.c_void_p), ("b_dec", C.c_void_p),
                ("out_scale", C.c_void_p)]


class HingeArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dim", C.c_int32), ("x_stride", C.c_int32), ("n_nodes", C.c_int64),
                ("node_mask", C.c_void_p), ("particle_id", C.c_void_p), ("r_emb", C.c_float), ("p", C.c_float),
                ("repulsive", C.c_int32), ("_pad", C.c_int32)]


_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "gnntrk_version": (C.c_int, []),
    "gnntrk_last_error": (C.c_char_p, []),
    "gnntrk_device_cu_count": (C.c_int, []),
    "gnntrk_graph_index_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "gnntrk_graph_index_build": (C.c_int, [_P, C.POINTER(GraphIndex), _P, C.c_size_t, _P]),
    "gnntrk_graph_index_build_ex": (C.c_int, [_P, C.POINTER(GraphIndex), _P, C.c_size_t, C.c_int32, _P]),
    "gnntrk_graph_index_workspace_bytes_carry": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "gnntrk_graph_index_build_carry": (C.c_int, [_P, C.POINTER(GraphIndex), C.POINTER(GraphIndexCarry), _P, C.c_size_t,
                                                 C.c_int32, _P]),
    "gnntrk_graph_index_place": (C.c_int, [C.POINTER(GraphIndex), C.c_int64, C.c_int64, C.POINTER(GraphIndex), _P, _P, _P, _P,
                                           _P, _P, _P, _P, _P]),
    "gnntrk_node_order_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnntrk_node_order": (C.c_int, [_P, C.c_int64, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    "gnntrk_bce_csr": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    "gnntrk_mlp_forward": (C.c_int, [C.POINTER(MlpFwdArgs), _P]),
    "gnntrk_rows_to_bf16": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P, C.c_int32, _P]),
    "gnntrk_segment_sum_bf16": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P, C.c_int32,
                                          _P]),
    "gnntrk_segment_sum_bf16_add": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P, C.c_int32, _P,
                                              C.c_int32, _P]),
    "gnntrk_permute_rows_bf16": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P, C.c_int32,
                                           C.c_int32, _P]),
    "gnntrk_mlp_forward_bf16": (C.c_int, [C.POINTER(MlpFwdArgs), _P]),
    "gnntrk_mlp_backward_workspace_bytes": (C.c_size_t, [C.POINTER(Mlp)]),
    "gnntrk_mlp_backward_bf16_workspace_bytes": (C.c_size_t, [C.POINTER(Mlp)]),
    "gnntrk_mlp_backward_bf16_max_terms": (C.c_int, [C.POINTER(MlpBwdArgs)]),
    "gnntrk_mlp_backward_bf16_can_fold": (C.c_int, [C.POINTER(MlpBwdArgs)]),
    "gnntrk_fold_finish_bf16": (C.c_int, [_P, C.c_int32, C.c_int64, _P, C.c_int64, _P, C.c_int32, _P]),
    "gnntrk_mlp_backward_bf16": (C.c_int, [C.POINTER(MlpBwdArgs), _P, C.c_size_t, _P]),
    "gnntrk_mlp_forward_bf16_kernel_name": (C.c_int, [C.POINTER(MlpFwdArgs), C.c_char_p, C.c_size_t]),
    "gnntrk_mlp_backward_bf16_kernel_name": (C.c_int, [C.POINTER(MlpBwdArgs), C.c_char_p, C.c_size_t]),
    "gnntrk_mlp_kernel_name": (C.c_int, [C.POINTER(Mlp), C.c_int32, C.POINTER(Seg), C.c_int32,
                                        C.c_char_p, C.c_size_t]),
    "gnntrk_mlp_backward": (C.c_int, [C.POINTER(MlpBwdArgs), _P, C.c_size_t, _P]),
    "gnntrk_segment_sum": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P,
                                     C.c_int32, C.c_int32, _P]),
    "gnntrk_permute_rows": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P, C.c_int32,
                                      C.c_int32, _P]),
    "gnntrk_axpby": (C.c_int, [C.c_float, _P, C.c_float, _P, _P, _P, C.c_int64, _P]),
    "gnntrk_bce_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnntrk_bce_forward": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_int64, _P, _P, C.c_size_t,
                                     _P]),
    "gnntrk_bce_backward": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_int64, _P, _P, _P]),
    "gnntrk_edge_targets_csr": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_float, C.c_int64, _P, _P]),
    "gnntrk_focal_forward": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                       C.c_int64, _P, _P, C.c_size_t, _P]),
    "gnntrk_focal_backward": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                        C.c_int64, _P, _P, _P]),
    "gnntrk_knn_search": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                    _P, _P, _P]),
    "gnntrk_knn_search_batched": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                            _P, C.c_int32, _P, _P, _P]),
    "gnntrk_knn_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "gnntrk_knn_search_ws": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, C.c_int32,
                                       _P, _P, _P, C.c_size_t, C.c_int32, _P]),
    "gnntrk_knn_emit": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int64, _P]),
    "gnntrk_knn_emit_prefix": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P]),
    "gnntrk_edge_labels": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "gnntrk_edge_features": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P, _P]),
    "gnntrk_compact_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnntrk_threshold_compact": (C.c_int, [_P, C.c_int64, C.c_float, _P, _P, _P, _P, C.c_size_t, _P]),
    "gnntrk_connected_nodes": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "gnntrk_radius_count": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_double, _P, _P, _P]),
    "gnntrk_radius_points_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "gnntrk_radius_edges_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnntrk_radius_count_ws": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, C.c_size_t,
                                         C.c_int32, _P]),
    "gnntrk_radius_fill_ws": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_double, _P, C.c_int64, _P, _P, _P,
                                        C.c_size_t, _P, C.c_size_t, C.c_int32, _P]),
    "gnntrk_radius_fill": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, _P]),
    "gnntrk_dbscan_init": (C.c_int, [_P, _P, C.c_int64, C.c_double, C.c_int32, _P, _P, _P]),
    "gnntrk_dbscan_propagate": (C.c_int, [_P, _P, _P, C.c_int64, C.c_double, _P, _P, C.c_int32, _P, _P]),
    "gnntrk_dbscan_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnntrk_dbscan_labels": (C.c_int, [_P, _P, _P, C.c_int64, C.c_double, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "gnntrk_good_node_mask": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, _P, _P]),
    "gnntrk_oc_select_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnntrk_oc_select_cps": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P,
                                       C.c_size_t, _P]),
    "gnntrk_oc_forward_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnntrk_oc_forward": (C.c_int, [C.POINTER(OcArgs), _P, _P, C.c_size_t, _P]),
    "gnntrk_oc_spatial_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "gnntrk_oc_forward_spatial": (C.c_int, [C.POINTER(OcArgs), _P, _P, C.c_size_t, _P]),
    "gnntrk_oc_backward_spatial": (C.c_int, [C.POINTER(OcArgs), _P, _P, _P, _P, C.c_int64, _P, C.c_size_t, _P]),
    "gnntrk_oc_backward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "gnntrk_oc_backward": (C.c_int, [C.POINTER(OcArgs), _P, _P, _P, _P, C.c_int64, _P, C.c_size_t, _P]),
    "gnntrk_mlp_wide_hidden_pad": (C.c_int32, [C.c_int32]),
    "gnntrk_mlp_wide_forward_workspace_bytes": (C.c_size_t, [C.POINTER(Mlp)]),
    "gnntrk_mlp_forward_wide": (C.c_int, [C.POINTER(MlpFwdArgs), _P, _P, C.c_size_t, _P]),
    "gnntrk_mlp_wide_backward_workspace_bytes": (C.c_size_t, [C.POINTER(Mlp), C.c_int64]),
    "gnntrk_mlp_backward_wide": (C.c_int, [C.POINTER(MlpBwdArgs), _P, _P, C.c_int32, _P, C.c_size_t, _P]),
    "gnntrk_resfcnn_hidden_pad": (C.c_int32, [C.c_int32]),
    "gnntrk_resfcnn_forward_workspace_bytes": (C.c_size_t, [C.POINTER(ResFcnn)]),
    "gnntrk_resfcnn_forward": (C.c_int, [C.POINTER(ResFcnn), _P, C.c_int32, C.c_int64, _P, C.c_int32, _P, _P,
                                         C.c_size_t, _P]),
    "gnntrk_resfcnn_backward_workspace_bytes": (C.c_size_t, [C.POINTER(ResFcnn), C.c_int64]),
    "gnntrk_resfcnn_backward": (C.c_int, [C.POINTER(ResFcnn), _P, C.c_int32, C.c_int64, _P, _P, C.c_int32, _P,
                                          C.c_int32, _P, C.c_int32, C.POINTER(ResFcnnGrads), C.c_int32, _P,
                                          C.c_size_t, _P]),
    "gnntrk_hinge_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "gnntrk_hinge_forward": (C.c_int, [C.POINTER(HingeArgs), _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    "gnntrk_hinge_backward": (C.c_int, [C.POINTER(HingeArgs), C.POINTER(GraphIndex), _P, _P, _P, C.c_int32,
                                        C.c_int32, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


ABI_VERSION = 600   # include/gnntrk.h: GNNTRK_VERSION the ctypes table below was written for


def bind(lib: C.CDLL) -> C.CDLL:
    """Attach restype/argtypes for every symbol of include/gnntrk.h."""
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.gnntrk_version()
    if got != ABI_VERSION:  # a stale library would be called with the wrong argument lists
        raise RuntimeError(f"libgnntrk.so reports ABI version {got}, this package binds version {ABI_VERSION}: "
                           "rebuild it (python -m gnn_tracking_amd._build)")
    return lib


_lib: C.CDLL | None = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load libgnntrk.so (building it in-tree first if absent).  Raises loudly."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = pathlib.Path(os.environ.get("GNNTRK_LIB", LIB_PATH))
        if "GNNTRK_LIB" not in os.environ:
            from . import _build

            # stamp-based and cheap when nothing changed: a library older than its sources
            # (edited csrc/, include/ or flags) is rebuilt instead of being loaded stale.
            # Without a compiler an existing library is used as it is.
            if _build.have_hipcc():
                _build.build_lib()
            elif not path.exists():
                raise RuntimeError(f"gnn_tracking_amd: {path} is missing and hipcc was not found; "
                                   "the package has no CPU fallback")
        try:
            lib = C.CDLL(str(path))
        except OSError as e:
            raise RuntimeError(
                f"gnn_tracking_amd: cannot load the HIP extension {path}: {e}. "
                "The package has no CPU fallback; build it with "
                "`python -m gnn_tracking_amd._build` on a ROCm machine.") from e
        _lib = bind(lib)
        return _lib


def require_device(*tensors) -> None:
    """Every tensor handed to the library must live in GPU memory (no CPU path)."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "gnn_tracking_amd: expected a HIP/CUDA tensor, got device "
                f"{t.device}. This package has no CPU implementation; move the data "
                "to the GPU (the CPU reference lives in oracle/ and is test-only).")


_DEBUG_SYNC = bool(os.environ.get("GNNTRK_DEBUG_SYNC"))


def check(rc: int, lib: C.CDLL | None = None) -> None:
    """Map a C return code to the reference's Python error conventions."""
    if rc == 0:
        if _DEBUG_SYNC:  # debugging aid: surface asynchronous faults at the offending call
            import sys
            import torch

            fr = sys._getframe(1)
            print(f"[gnntrk] sync after {fr.f_code.co_name}:{fr.f_lineno}", flush=True)
            torch.cuda.synchronize()
        return
    lib = lib or load()
    msg = (lib.gnntrk_last_error() or b"").decode(errors="replace")
    if rc == 3:
        # utils/oom.py:12-18 matches on "out of memory"
        raise RuntimeError(f"HIP out of memory: {msg}")
    if rc == 4:
        raise NotImplementedError(f"gnntrk: {msg}")
    if rc == 1:
        raise ValueError(f"gnntrk: {msg}")
    raise RuntimeError(f"gnntrk: {msg} (code {rc})")


def make_mlp(weights, biases, in_dim: int, hidden: int, out_dim: int) -> Mlp:
    """weights/biases: sequences of raw addresses (0/None = absent)."""
    m = Mlp()
    m.n_layers, m.in_dim, m.hidden, m.out_dim = len(weights), in_dim, hidden, out_dim
    for i, w in enumerate(weights):
        m.W[i] = w
        m.b[i] = (biases[i] or None) if biases is not None else None
    return m
