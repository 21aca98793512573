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
    assert lib.gnntrk_version() == 500
    # argument validation happens on the host, before any launch
    rc = lib.gnntrk_mlp_forward(None, None)
    assert rc == 1 and b"NULL" in lib.gnntrk_last_error()


def test_struct_layout_matches_header():
    from gnn_tracking_amd import _capi

    assert ctypes.sizeof(_capi.Seg) == 32
    assert ctypes.sizeof(_capi.Mlp) == 64
    assert ctypes.sizeof(_capi.MlpFwdArgs) == 448
    assert _capi.MlpFwdArgs.n_rows.offset == 392
    assert ctypes.sizeof(_capi.GraphIndex) == 72
    assert ctypes.sizeof(_capi.GraphIndexCarry) == 48
    assert ctypes.sizeof(_capi.MlpBwdArgs) == 784
    assert ctypes.sizeof(_capi.ResFcnn) == 8 * (5 + 2 * _capi.RESFCNN_MAX_HIDDEN) + 32
    assert ctypes.sizeof(_capi.ResFcnnGrads) == 8 * (5 + 2 * _capi.RESFCNN_MAX_HIDDEN)
    assert ctypes.sizeof(_capi.HingeArgs) == 56 and _capi.HingeArgs.r_emb.offset == 40


def test_product_path_refuses_cpu_tensors():
    import torch

    import gnn_tracking_amd as G

    with pytest.raises(RuntimeError, match="no CPU implementation"):
        G.MLP(4, 2, 8)(torch.zeros(3, 4))
