"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/gnntrk.h declares (no compute calls: there is no GPU in this container)."""

import ctypes
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib_path():
    from gnn_tracking_amd import _build

    return _build.build_lib()


def test_header_symbols_are_exported(lib_path):
    from gnn_tracking_amd import _capi

    hdr = (ROOT / "include" / "gnntrk.h").read_text()
    declared = set(re.findall(r"\b(gnntrk_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(str(lib_path))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in gnntrk.h but not exported"
    assert declared == set(_capi.EXPORTED_SYMBOLS), "ctypes table out of sync with gnntrk.h"


def test_version_and_error_channel(lib_path):
    from gnn_tracking_amd import _capi

    lib = _capi.bind(ctypes.CDLL(str(lib_path)))
    assert lib.gnntrk_version() == 600
    # argument validation happens on the host, before any launch
    rc = lib.gnntrk_mlp_forward(None, None)
    assert rc == 1 and b"NULL" in lib.gnntrk_last_error()


def test_round5_entry_points_validate_on_the_host(lib_path):
    """gnntrk_node_order / gnntrk_graph_index_place refuse bad arguments before any launch (include/gnntrk.h)."""
    from gnn_tracking_amd import _capi

    lib = _capi.bind(ctypes.CDLL(str(lib_path)))
    assert lib.gnntrk_node_order_workspace_bytes(1000) >= 1000 * 24
    assert lib.gnntrk_node_order(None, 1, None, 0, 0, None, None, None, 0, None) == 0      # nothing to order
    assert lib.gnntrk_node_order(None, 1, None, 0, 10, None, None, None, 0, None) == 1 and b"NULL" in lib.gnntrk_last_error()
    buf = (ctypes.c_float * 16)()
    out = (ctypes.c_int32 * 16)()
    rc = lib.gnntrk_node_order(buf, 1, None, 0, 16, out, out, buf, 8, None)
    assert rc == 1 and b"workspace" in lib.gnntrk_last_error()
    assert lib.gnntrk_node_order(buf, 1, None, 0, 1 << 32, out, out, buf, 8, None) == 4   # GNNTRK_EUNSUPPORTED: sizes must fit int32
    assert lib.gnntrk_graph_index_place(None, 0, 0, None, *([None] * 8), None) == 1
    part, batch = _capi.GraphIndex(), _capi.GraphIndex()
    part.n_nodes, part.n_edges, batch.n_nodes, batch.n_edges = 10, 20, 15, 30
    rc = lib.gnntrk_graph_index_place(ctypes.byref(part), 6, 0, ctypes.byref(batch), *([None] * 8), None)
    assert rc == 1 and b"does not fit" in lib.gnntrk_last_error()
    rc = lib.gnntrk_graph_index_place(ctypes.byref(part), 0, 0, ctypes.byref(batch), out, None, *([None] * 6), None)
    assert rc == 1 and b"pairs" in lib.gnntrk_last_error()


def test_struct_layout_matches_header():
    from gnn_tracking_amd import _capi

    assert ctypes.sizeof(_capi.Seg) == 32
    assert ctypes.sizeof(_capi.Mlp) == 64
    assert ctypes.sizeof(_capi.MlpFwdArgs) == 448
    assert _capi.MlpFwdArgs.n_rows.offset == 392
    assert ctypes.sizeof(_capi.GraphIndex) == 72
    assert ctypes.sizeof(_capi.GraphIndexCarry) == 48
    assert ctypes.sizeof(_capi.MlpBwdArgs) == 808 and ctypes.sizeof(_capi.GFold) == 24
    assert ctypes.sizeof(_capi.ResFcnn) == 8 * (5 + 2 * _capi.RESFCNN_MAX_HIDDEN) + 32
    assert ctypes.sizeof(_capi.ResFcnnGrads) == 8 * (5 + 2 * _capi.RESFCNN_MAX_HIDDEN)
    assert ctypes.sizeof(_capi.HingeArgs) == 56 and _capi.HingeArgs.r_emb.offset == 40


def test_product_path_refuses_cpu_tensors():
    import torch

    import gnn_tracking_amd as G

    with pytest.raises(RuntimeError, match="no CPU implementation"):
        G.MLP(4, 2, 8)(torch.zeros(3, 4))
