"""Run the package's Python layer against the CPU wave64 emulator build of the kernel
sources (tests/emul).  TEST INFRASTRUCTURE ONLY: the product loader
(gnn_tracking_amd/_capi.py) knows nothing about the emulator; tests swap the library
handle and the device guard explicitly through this context manager."""

from __future__ import annotations

import contextlib
import ctypes
import importlib.util
import pathlib

HERE = pathlib.Path(__file__).resolve().parent


def _build(asan=False):
    spec = importlib.util.spec_from_file_location("build_emul", HERE / "emul" / "build_emul.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(asan=asan)


_EMUL = None


def emulator_lib():
    global _EMUL
    if _EMUL is None:
        from gnn_tracking_amd import _capi

        _EMUL = _capi.bind(ctypes.CDLL(str(_build())))
    return _EMUL


@contextlib.contextmanager
def emulated():
    from gnn_tracking_amd import _capi

    old_lib, old_guard = _capi._lib, _capi.require_device
    _capi._lib = emulator_lib()
    _capi.require_device = lambda *a: None
    try:
        yield
    finally:
        _capi._lib, _capi.require_device = old_lib, old_guard
