"""Hyper-parameter bookkeeping with the surface the reference modules rely on.

The reference mixes ``pytorch_lightning``'s ``HyperparametersMixin`` into every
``nn.Module`` (e.g. models/interaction_network.py:12,37) and reads ctor arguments
back through ``self.hparams.<name>``.  Lightning is not a dependency of this
package; this mixin provides the same three behaviours the hot path uses:
``save_hyperparameters()`` (collect the caller's ``__init__`` arguments),
``save_hyperparameters(ignore=[...])`` and ``save_hyperparameters({key: value})``,
plus attribute-style access on ``hparams``.  Sub-module serialisation follows
utils/lightning.py:18-80 (``{"class_path": ..., "init_args": ...}``).
"""

from __future__ import annotations

import importlib
import inspect
from typing import Any


class AttributeDict(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(f"Missing attribute '{key}'") from e

    def __setattr__(self, key, value):
        self[key] = value


class HyperparametersMixin:
    @property
    def hparams(self) -> AttributeDict:
        if "_hparams" not in self.__dict__:
            object.__setattr__(self, "_hparams", AttributeDict())
        return self.__dict__["_hparams"]

    def save_hyperparameters(self, *args: Any, ignore=None, **_unused) -> None:
        if args and isinstance(args[0], dict):
            self.hparams.update(args[0])
            return
        ignore = set([ignore] if isinstance(ignore, str) else (ignore or []))
        frame = inspect.currentframe().f_back
        while frame is not None and not (
            frame.f_code.co_name == "__init__" and frame.f_locals.get("self") is self
        ):
            frame = frame.f_back
        if frame is None:
            return
        info = inspect.getargvalues(frame)
        for name in info.args:
            if name != "self" and name not in ignore:
                self.hparams[name] = info.locals[name]
        if info.keywords:
            for k, v in info.locals[info.keywords].items():
                if k not in ignore:
                    self.hparams[k] = v


def get_object_from_path(path: str, init_args: dict | None = None) -> Any:
    """utils/lightning.py:83-94."""
    module_name, _, class_name = path.rpartition(".")
    if not module_name:
        raise ValueError("Please specify the full import path")
    obj = getattr(importlib.import_module(module_name), class_name)
    return obj(**init_args) if init_args is not None else obj


def obj_from_or_to_hparams(self: HyperparametersMixin, key: str, obj: Any) -> Any:
    """utils/lightning.py:66-80: dict with class_path/init_args -> instantiate;
    object with hparams -> record its class path and init args."""
    if isinstance(obj, dict) and "class_path" in obj and "init_args" in obj:
        self.save_hyperparameters({key: obj})
        return get_object_from_path(obj["class_path"], obj["init_args"])
    if isinstance(obj, (int, float, str, bool, list, tuple, dict)) or obj is None:
        self.save_hyperparameters({key: obj})
        return obj
    if hasattr(obj, "hparams"):
        assert key not in self.hparams
        self.save_hyperparameters({key: {
            "class_path": obj.__class__.__module__ + "." + obj.__class__.__name__,
            "init_args": dict(obj.hparams),
        }})
    return obj


def assert_feat_dim(feat_vec, dim: int) -> None:
    """utils/asserts.py:4-7."""
    assert feat_vec.shape[-1] == dim, (
        f"Expected feature dimension {dim}, got {feat_vec.shape[-1]}")
