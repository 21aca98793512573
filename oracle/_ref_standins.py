"""Stand-ins for the third-party packages the reference imports but this image lacks.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/make_golden.py`` (run in the build
container, where ``/root/reference`` exists) so that the reference's *own*
Python files for the hot path can be imported and executed to (a) validate the
CPU restatement in ``oracle/ref_cpu.py`` and (b) dump golden vectors into
``tests/golden``.  Nothing here travels into the product package and nothing in
``gnn_tracking_amd`` imports it.

The stand-ins follow the *documented* semantics of the packages they replace
(SURVEY.md section 8c):

* ``pytorch_lightning.core.mixins.hparams_mixin.HyperparametersMixin`` -
  ``save_hyperparameters()`` collects the caller's constructor arguments into an
  attribute dict ``self.hparams``.
* ``torch_geometric.nn.MessagePassing`` - ``propagate`` for
  ``flow="source_to_target"``, ``aggr="add"``: ``x_j = x[edge_index[0]]``,
  ``x_i = x[edge_index[1]]``, messages are summed onto ``edge_index[1]``.
* ``torch_geometric.data.Data`` - attribute bag with ``num_nodes``,
  ``edge_subgraph`` and ``subgraph``.
* ``torch_cluster.knn_graph / radius_graph / knn`` - exact brute force; edges are
  ``[neighbour, query]`` grouped by query, ascending distance, self excluded by
  index; ``radius_graph`` keeps ``d < r`` with a per-query cap.
"""

from __future__ import annotations

import inspect
import logging
import sys
import types

import torch


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class HyperparametersMixin:
    """save_hyperparameters(): read the calling frame's ctor args."""

    @property
    def hparams(self):
        if not hasattr(self, "_hparams_store"):
            object.__setattr__(self, "_hparams_store", _AttrDict())
        return self._hparams_store

    def save_hyperparameters(self, *args, ignore=None, **_kw):
        if args and isinstance(args[0], dict):
            self.hparams.update(args[0])
            return
        frame = inspect.currentframe().f_back
        # climb to the __init__ frame of `self`
        while frame is not None and not (
            frame.f_code.co_name == "__init__" and frame.f_locals.get("self") is self
        ):
            frame = frame.f_back
        if frame is None:  # pragma: no cover
            return
        argvals = inspect.getargvalues(frame)
        ignore = set(ignore or [])
        for name in argvals.args:
            if name == "self" or name in ignore:
                continue
            self.hparams[name] = argvals.locals[name]
        if argvals.keywords:
            for k, v in argvals.locals[argvals.keywords].items():
                if k not in ignore:
                    self.hparams[k] = v


class Data:
    """Attribute bag with the small part of the PyG ``Data`` API the path uses."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __getitem__(self, k):
        return getattr(self, k)

    def __contains__(self, k):
        return k in self.__dict__

    def __getattr__(self, k):
        # PyG ``Data`` declares x / edge_index / edge_attr / y / pos / batch ... as properties that
        # return None when the attribute was never set (training/tc.py:61 reads ``data.batch`` of a
        # graph built by MLGraphConstruction, which has none)
        if k in ("batch", "pos", "edge_weight", "edge_attr", "y", "x", "edge_index"):
            return None
        raise AttributeError(k)

    @property
    def num_nodes(self):
        return self.x.shape[0]

    @property
    def num_edges(self):
        return self.edge_index.shape[1]

    def _is_edge_attr(self, k, v):
        return (
            torch.is_tensor(v)
            and k != "edge_index"
            and v.dim() >= 1
            and v.shape[0] == self.num_edges
            and (k.startswith("edge") or k in ("y", "ec_edge_embedding"))
        )

    def edge_subgraph(self, mask):
        out = Data()
        E = self.num_edges
        for k in self.keys():
            v = getattr(self, k)
            if k == "edge_index":
                out.edge_index = v[:, mask]
            elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == E and (
                k.startswith("edge") or k in ("y", "ec_edge_embedding")
            ):
                setattr(out, k, v[mask])
            else:
                setattr(out, k, v)
        return out

    def subgraph(self, subset):
        N = self.num_nodes
        if subset.dtype == torch.bool:
            node_mask = subset
        else:
            node_mask = torch.zeros(N, dtype=torch.bool)
            node_mask[subset] = True
        relabel = torch.full((N,), -1, dtype=torch.long)
        relabel[node_mask] = torch.arange(int(node_mask.sum()))
        ei = self.edge_index
        emask = node_mask[ei[0]] & node_mask[ei[1]]
        out = Data()
        E = self.num_edges
        for k in self.keys():
            v = getattr(self, k)
            if k == "edge_index":
                out.edge_index = relabel[ei[:, emask]]
            elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == E and (
                k.startswith("edge") or k in ("y", "ec_edge_embedding")
            ):
                setattr(out, k, v[emask])
            elif torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == N:
                setattr(out, k, v[node_mask])
            else:
                setattr(out, k, v)
        return out


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", **_kw):
        super().__init__()
        assert aggr == "add" and flow == "source_to_target"

    def propagate(self, edge_index, size=None, **kwargs):
        x = kwargs["x"]
        x_j = x.index_select(0, edge_index[0])
        x_i = x.index_select(0, edge_index[1])
        msg_kwargs = {k: v for k, v in kwargs.items() if k != "x"}
        msg = self.message(x_i=x_i, x_j=x_j, **msg_kwargs)
        out = torch.zeros(x.shape[0], msg.shape[1], dtype=msg.dtype)
        out = out.scatter_add(0, edge_index[1].view(-1, 1).expand_as(msg), msg)
        return self.update(out, x=x)


def index_to_mask(index, size=None):
    size = int(index.max()) + 1 if size is None else size
    mask = torch.zeros(size, dtype=torch.bool)
    mask[index] = True
    return mask


def _bf_neighbours(x, y, k, batch_x=None, batch_y=None, loop=False, r=None, same=True):
    """For every row of y: the k nearest rows of x (ascending; ties -> lower index)."""
    d = torch.cdist(y.double(), x.double())  # exact enough for the small goldens
    if batch_x is not None:
        d = d.masked_fill(batch_y.view(-1, 1) != batch_x.view(1, -1), float("inf"))
    if same and not loop:
        d.fill_diagonal_(float("inf"))
    kk = min(k, x.shape[0])
    dist, idx = torch.sort(d, dim=1, stable=True)
    dist, idx = dist[:, :kk], idx[:, :kk]
    ok = torch.isfinite(dist)
    if r is not None:
        ok &= dist < r
    q = torch.arange(y.shape[0]).view(-1, 1).expand_as(idx)
    return torch.stack([idx[ok], q[ok]])


def knn_graph(x, k, batch=None, loop=False, flow="source_to_target", **_kw):
    assert flow == "source_to_target"
    return _bf_neighbours(x, x, k, batch, batch, loop=loop)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32,
                 flow="source_to_target", **_kw):
    assert flow == "source_to_target"
    return _bf_neighbours(x, x, max_num_neighbors, batch, batch, loop=loop, r=r)


def knn(x, y, k, batch_x=None, batch_y=None, **_kw):
    e = _bf_neighbours(x, y, k, batch_x, batch_y, same=False)
    return torch.stack([e[1], e[0]])


def install() -> None:
    """Inject the stand-in modules into ``sys.modules`` (idempotent)."""
    if "pytorch_lightning" in sys.modules and getattr(
        sys.modules["pytorch_lightning"], "_gnntrk_standin", False
    ):
        return
    sys.dont_write_bytecode = True

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    class LightningModule(torch.nn.Module, HyperparametersMixin):
        pass

    pl = mod("pytorch_lightning", LightningModule=LightningModule,
             LightningDataModule=_Dummy, Callback=_Dummy, Trainer=_Dummy,
             _gnntrk_standin=True)
    mod("pytorch_lightning.core")
    mod("pytorch_lightning.core.mixins")
    mod("pytorch_lightning.core.mixins.hparams_mixin",
        HyperparametersMixin=HyperparametersMixin)
    mod("pytorch_lightning.callbacks", ProgressBar=_Dummy, Callback=_Dummy)
    mod("pytorch_lightning.cli", LightningCLI=_Dummy, OptimizerCallable=object,
        LRSchedulerCallable=object)
    mod("pytorch_lightning.loggers", WandbLogger=_Dummy, TensorBoardLogger=_Dummy)
    mod("pytorch_lightning.utilities")
    pl.core = sys.modules["pytorch_lightning.core"]
    pl.callbacks = sys.modules["pytorch_lightning.callbacks"]
    pl.cli = sys.modules["pytorch_lightning.cli"]
    pl.loggers = sys.modules["pytorch_lightning.loggers"]

    # utils/nomenclature.py:3,28 (run names; reached through training/ml.py -> k_scanner -> cluster_metrics)
    mod("coolname", generate_slug=lambda n=3: "-".join(["standin"] * int(n)))

    class Metric(torch.nn.Module):
        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, default)

    mod("torchmetrics", Metric=Metric)
    mod("torchmetrics.classification", BinaryAUROC=_Dummy)

    cl = mod("colorlog", getLogger=logging.getLogger, StreamHandler=logging.StreamHandler)

    class ColoredFormatter(logging.Formatter):
        def __init__(self, fmt=None, log_colors=None, datefmt=None, **_k):
            super().__init__((fmt or "%(message)s").replace("%(log_color)s", ""), datefmt)

    cl.ColoredFormatter = ColoredFormatter

    tg = mod("torch_geometric")
    tgd = mod("torch_geometric.data", Data=Data, Batch=_Dummy)
    tgn = mod("torch_geometric.nn", MessagePassing=MessagePassing)
    mod("torch_geometric.nn.conv", MessagePassing=MessagePassing)
    mod("torch_geometric.utils", index_to_mask=index_to_mask)
    mod("torch_geometric.typing", OptTensor=object, PairOptTensor=object,
        PairTensor=object)
    mod("torch_geometric.loader", DataLoader=_Dummy)
    tg.data, tg.nn = tgd, tgn

    mod("torch_cluster", knn=knn, knn_graph=knn_graph, radius_graph=radius_graph)
