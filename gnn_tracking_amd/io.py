"""Reading the reference's on-disk graphs without PyG.

The reference stores one ``torch_geometric.data.Data`` per event with ``torch.save``
(graph_construction/graph_builder.py:396-455: ``x`` f32[N,14], ``edge_index`` i64[2,E],
``edge_attr`` f32[E,4], ``y``, ``pt``, ``particle_id`` i64, ``reconstructable``, ``sector``,
``evtid``, ``s``, ``eta``, ``layer``) and reads them back in utils/loading.py:17-113.
Such a file is a pickle that references PyG classes; here it is opened with a RESTRICTED
unpickler: tensors and plain containers are rebuilt, the PyG container classes are mapped to
inert stand-ins whose ``_store._mapping`` dict carries the fields, everything else is refused
(no arbitrary code execution from a data file).

``PrefetchLoader`` overlaps disk reads + host-to-device copies of the next graphs with the
training step: a background thread fills pinned host buffers, the copy runs on a side stream.
"""

from __future__ import annotations

import collections
import io as _io
import pathlib
import pickle
import queue
import threading
from typing import Iterable, Iterator, Sequence

import torch

from .data import Data, collate

_PYG_CONTAINERS = {
    ("torch_geometric.data.data", "Data"), ("torch_geometric.data.data", "DataEdgeAttr"),
    ("torch_geometric.data.data", "DataTensorAttr"), ("torch_geometric.data.storage", "GlobalStorage"),
    ("torch_geometric.data.storage", "BaseStorage"), ("torch_geometric.data.storage", "NodeStorage"),
    ("torch_geometric.data.storage", "EdgeStorage"),
}
_ALLOWED = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "dict"),
    ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "int"),
    ("builtins", "float"), ("builtins", "str"), ("builtins", "bool"), ("builtins", "slice"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"),
    ("torch._utils", "_rebuild_parameter"), ("torch", "Size"), ("torch", "device"),
    ("torch.serialization", "_get_layout"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"), ("numpy", "dtype"),
}


class _Inert:
    """Stand-in for a PyG container: keeps whatever state the pickle assigns."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {"_state": st})


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _PYG_CONTAINERS:
            return type(name, (_Inert,), {})
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module == "torch" and (name.endswith("Storage") or name in (
                "float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8",
                "bool")):
            return getattr(torch, name)
        raise pickle.UnpicklingError(f"graph file references {module}.{name}: not a tensor container, refused")


class _RestrictedPickle:
    """Duck-typed ``pickle_module`` for ``torch.load``."""

    __name__ = "gnn_tracking_amd.io.restricted_pickle"
    Unpickler = _RestrictedUnpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kw):
        return _RestrictedUnpickler(f, **kw).load()


def _fields(obj) -> dict:
    if isinstance(obj, dict):
        return obj
    store = obj.__dict__.get("_store", None)
    if store is not None:
        mapping = store.__dict__.get("_mapping", None)
        if isinstance(mapping, dict):
            return mapping
    return {k: v for k, v in obj.__dict__.items() if not k.startswith("_")}


def load_graph(path, device=None) -> Data:
    """Read one reference ``.pt`` graph into this package's ``Data`` (no PyG needed)."""
    with open(path, "rb") as f:
        payload = f.read()
    obj = torch.load(_io.BytesIO(payload), map_location="cpu", pickle_module=_RestrictedPickle,
                     weights_only=False)
    fields = {k: v for k, v in _fields(obj).items() if v is not None}
    if "x" not in fields or "edge_index" not in fields:
        raise ValueError(f"{path}: not a graph file (fields: {sorted(fields)})")
    d = Data(**fields)
    return d if device is None else d.to(device)


def renumber_nodes(data: Data, col: int | None = None) -> Data:
    """The hits of ONE event renumbered by a feature column of ``data.x`` (default: the azimuth column the edge
    classifier would sort by itself, ``locality.AUTO_COLUMN``): every node-level attribute is permuted, every
    ``*index*`` attribute relabelled, ``node_order_key`` records the column and ``node_perm`` (new id -> old id) maps
    node results back.  The reference keeps the hits in hit-table order (graph_construction/graph_builder.py:396-455,
    read back unchanged by utils/loading.py:17-113); its datasets are static, so a loader can pay for the geometric
    order ONCE per event (``GraphDataset(renumber=True)``) instead of the 1.4 ms per 64 M-edge step that
    ``ECForGraphTCN`` spends when a batch arrives in file order.  An event that carries
    ``node_order_key`` is not renumbered again in the step (``locality.order_column``).  The order is the stable sort
    of the key's order-preserving integer image; the in-step renumbering of a batch sorts by a QUANTISED image of the
    same key (ties stable), so the two numberings agree in locality, not id for id - results are equal up to
    summation order either way."""
    import copy

    from . import locality

    col = locality.AUTO_COLUMN if col is None else int(col)
    x = data.x
    if x.dim() != 2 or x.dtype != torch.float32 or not 0 <= col < x.shape[1]:
        raise ValueError("renumber_nodes: data.x must be fp32 [N, F] with the key column inside")
    if x.is_cuda:
        from . import ops
        perm, rank = (t.long() for t in ops.node_order(x.contiguous(), col, None))
    else:
        u = x[:, col].contiguous().view(torch.int32).long() & 0xffffffff
        perm = torch.argsort(torch.where(u >> 31 == 1, u ^ 0xffffffff, u ^ 0x80000000), stable=True)
        rank = torch.empty_like(perm)
        rank[perm] = torch.arange(perm.numel(), device=perm.device)
    out = copy.copy(data)
    for k in data.keys():
        v = getattr(data, k)
        if not torch.is_tensor(v):
            continue
        if "index" in k:
            setattr(out, k, rank[v])
        elif data.is_node_attr(k):
            setattr(out, k, v[perm])
    out.node_perm = perm
    out.node_order_key = col
    return out


class GraphDataset:
    """The ``*.pt`` files of one or several directories (utils/loading.py:17-113: sorted by
    name, optional ``start``/``stop`` slice and sector filter).  ``renumber``: every graph goes through
    ``renumber_nodes`` when it is read (True: the default key column, or a column number)."""

    def __init__(self, in_dirs: str | Sequence[str], *, start: int = 0, stop: int | None = None,
                 sector: int | None = None, renumber: bool | int = False):
        self.renumber = renumber
        dirs = [in_dirs] if isinstance(in_dirs, (str, pathlib.Path)) else list(in_dirs)
        files: list[pathlib.Path] = []
        for d in dirs:
            files += sorted(pathlib.Path(d).glob("*.pt"))
        if sector is not None:
            files = [f for f in files if f.stem.endswith(f"_s{sector}")]
        self.files = files[start:stop]

    def __len__(self) -> int:
        return len(self.files)

    def __getitem__(self, i: int) -> Data:
        g = load_graph(self.files[i])
        if self.renumber is not False:
            g = renumber_nodes(g, None if self.renumber is True else int(self.renumber))
        return g


class ResidentDataset:
    """A static dataset kept ON THE DEVICE, with the graph index of every event built ONCE.

    The reference's datasets do not change between epochs (utils/loading.py:97-100: the same files every epoch,
    shuffled), and 288 GB of HBM hold thousands of 2 M-edge events (about 150 MB each with the index).  Every
    event is read, copied and indexed on first use - node order (``locality.py``), 1-byte labels and, for bf16
    storage, the edge features carried into CSR order; ``batches`` collates events as the reference's DataLoader
    does and PLACES their indices into the batch's arrays (``ops.place_graph_indices``: one streaming pass per
    event, the arrays identical to building the index of the collated list) instead of sorting 64 M edges in every
    step.  ``ECForGraphTCN`` finds the placed index through ``ops.placed_graph_index``.

        ds = ResidentDataset(GraphDataset(dirs), "cuda:0")
        for epoch in range(n):
            for batch in ds.batches(32, shuffle=True, seed=epoch):
                module.optimisation_step(batch)
    """

    def __init__(self, events, device, *, bf16: bool = True, order: bool | int = True):
        """``events``: a sequence of ``Data`` (``GraphDataset``, a list); ``bf16``: carry the four fp32 edge features
        as bf16 rows (what ``bf16_storage`` reads; fp32 runs gather ``edge_attr`` through the permutation);
        ``order``: True = follow the node-order policy of ``locality.py`` (column ``AUTO_COLUMN`` unless the policy is
        off - whatever the size: the sort is paid once), a column number, or False."""
        self.source, self.device = events, torch.device(device)
        self.bf16, self.order = bool(bf16), order
        self._events: list = [None] * len(events)
        self._parts: list = [None] * len(events)

    def __len__(self) -> int:
        return len(self._events)

    def _policy_column(self):
        """The column this dataset orders its events by (None: it does not)."""
        from . import locality

        if self.order is False or (self.order is True and locality.mode() == "off"):
            return None
        return int(self.order) if self.order is not True else (
            locality.AUTO_COLUMN if locality.mode() == "auto" else int(locality.mode()))

    def _key_column(self, e: Data):
        col = self._policy_column()
        if col is None or "node_order_key" in e:
            return None
        x = e.x
        return col if x.dim() == 2 and x.dtype == torch.float32 and 0 <= col < x.shape[1] and x.shape[0] >= 2 else None

    def event(self, i: int):
        """``(data, index)`` of event ``i`` on the device (read and indexed on first use)."""
        if self._events[i] is None:
            from . import ops

            e = self.source[i].to(self.device)
            col = self._key_column(e)
            y, ea = getattr(e, "y", None), getattr(e, "edge_attr", None)
            part = ops.graph_index(e.edge_index, e.num_nodes, cache=False,
                                   carry_label=y if torch.is_tensor(y) else None,
                                   carry_rows=ea if self.bf16 and torch.is_tensor(ea) and ea.dim() == 2 and ea.shape[1] == 4 else None,
                                   order_by=None if col is None else (e.x, col, None))
            self._events[i], self._parts[i] = e, part
        return self._events[i], self._parts[i]

    def batches(self, batch_size: int = 1, *, shuffle: bool = False, seed: int = 0) -> Iterator[Data]:
        """Collated batches of ``batch_size`` events with their graph index placed (see the class)."""
        from . import ops

        idx = list(range(len(self)))
        if shuffle:
            idx = torch.randperm(len(idx), generator=torch.Generator().manual_seed(seed)).tolist()
        for s in range(0, len(idx), int(batch_size)):
            evs = [self.event(i) for i in idx[s:s + int(batch_size)]]
            batch = collate([e for e, _ in evs])
            # (events the loader renumbered when it read them keep their order: nothing to place through)
            oc = self._policy_column() if any("node_order_key" not in e for e, _ in evs) else None
            ops.place_graph_indices([p for _, p in evs], batch, order_col=oc)
            yield batch


class PrefetchLoader:
    """Iterate over batches of ``batch_size`` collated graphs on ``device``.  A background
    thread reads and collates the next ``depth`` batches into pinned memory; the
    host-to-device copy is issued on a side stream and joined when the batch is handed out."""

    def __init__(self, dataset, *, batch_size: int = 1, device=None, depth: int = 2, shuffle: bool = False,
                 seed: int = 0, build_index: bool = False):
        """``build_index``: also build the graph index of every staged batch on the side stream
        (``ops.prefetch_graph_index``: it depends on ``edge_index`` only), so the step that
        consumes the batch joins a finished index instead of building it."""
        self.build_index = bool(build_index)
        self.dataset, self.batch_size, self.depth = dataset, int(batch_size), max(1, int(depth))
        self.device = torch.device(device) if device is not None else None
        self.shuffle, self.seed, self._epoch = shuffle, seed, 0

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _order(self) -> list[int]:
        idx = list(range(len(self.dataset)))
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self._epoch)
            idx = torch.randperm(len(idx), generator=g).tolist()
        return idx

    def __iter__(self) -> Iterator[Data]:
        order = self._order()
        self._epoch += 1
        q: queue.Queue = queue.Queue(maxsize=self.depth)
        cuda = self.device is not None and self.device.type == "cuda"
        side = torch.cuda.Stream(self.device) if cuda else None

        def produce():
            try:
                for s in range(0, len(order), self.batch_size):
                    batch = collate([self.dataset[i] for i in order[s:s + self.batch_size]])
                    if cuda:
                        batch = batch._map(lambda t: t.pin_memory())
                        with torch.cuda.stream(side):
                            dev_batch = batch.to(self.device, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(side)
                        if self.build_index:
                            from . import ops
                            from . import locality
                            # (the same predicate as the step's: events renumbered by the policy's column when they
                            #  were read keep their order - no x, no renumbering in the build)
                            ops.prefetch_graph_index(dev_batch.edge_index, dev_batch.num_nodes, side,
                                                     x=None if locality.order_column(dev_batch) is None else dev_batch.x,
                                                     batch=getattr(dev_batch, "batch", None))
                        q.put((dev_batch, ev, batch))  # keep the pinned source alive until consumed
                    else:
                        q.put((batch if self.device is None else batch.to(self.device), None, None))
                q.put(None)
            except BaseException as e:  # surface loader errors in the consumer
                q.put(e)

        threading.Thread(target=produce, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            batch, ev, _pinned = item
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                # the device tensors were allocated on the side stream: tell the caching
                # allocator that the consumer's stream uses them, or the producer's next
                # copy could be handed the same blocks while kernels queued here still read them
                batch._map(lambda t: (t.record_stream(cur), t)[1])
            yield batch
