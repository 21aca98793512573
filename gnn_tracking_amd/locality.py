"""Node-order policy: renumber the hits of every event by a geometric key before the message passing.

The reference keeps an event's hits in the order of the hit table (graph_construction/graph_builder.py:
396-455; utils/loading.py:17-113 hands the graphs on unchanged), which is unrelated to the geometry:
``x[edge_index[0]]`` (models/interaction_network.py:67) then gathers random 16-byte rows - 64-byte sectors
fetched for 16-byte rows, and per-edge source gradients scattered the same way.  Edges of tracking graphs
join hits of neighbouring azimuth (graph_builder.py: the phi-slope cut), so sorting every event's hits by
the azimuth column of ``data.x`` (column 1: ``r, phi, z, ...``) makes both local.

The renumbering is internal to ``ECForGraphTCN``: the graph index is built in the new numbering
(``ops.graph_index(order_by=...)``: a counting sort of the nodes on a quantised key, ids translated before the index
build reads them), the node encoder gathers ``x`` through the permutation, and ``node_embedding`` is handed back
in the caller's order.  Any key gives the same results up to summation order; the key only decides speed.

The key should be CONTINUOUS (an angle, a coordinate): the in-step renumbering quantises it to 65 536 levels per event
and counting-sorts (event, level) pairs; a column with a handful of distinct values (a layer number) puts thousands of
hits on one level, which the sort's bucket stage ranks through its slow path (correct, about 5 us per 1 000 hits of
such a level) - and buys no locality anyway.

    GNNTRK_NODE_ORDER = auto (default) | off | <column of data.x>
    with gnn_tracking_amd.node_order("off"): ...     # or "auto", or a column number
"""

from __future__ import annotations

import contextlib
import os

_MODE = os.environ.get("GNNTRK_NODE_ORDER", "auto")
#: below this many nodes the node rows of a batch sit in the caches whatever their order
MIN_NODES = int(os.environ.get("GNNTRK_NODE_ORDER_MIN", "65536"))
AUTO_COLUMN = 1   # phi of the reference's node features (graph_builder.py: r, phi, z, eta_rz, u, v, ...)


def _parse(mode) -> str:
    m = str(mode).lower()
    if m in ("off", "none", "no", "false"):
        return "off"
    if m == "auto":
        return "auto"
    int(m)   # (raises for anything that is not a column number)
    return m


_MODE = _parse(_MODE)


def mode() -> str:
    return _MODE


@contextlib.contextmanager
def node_order(mode, min_nodes: int | None = None):
    """Node-order policy inside the block: "off", "auto" (azimuth column from ``MIN_NODES`` nodes on) or a
    column number of ``data.x`` (applied at any size unless ``min_nodes`` says otherwise)."""
    global _MODE, MIN_NODES
    old = (_MODE, MIN_NODES)
    _MODE = _parse(mode)
    if min_nodes is not None:
        MIN_NODES = int(min_nodes)
    elif _MODE not in ("off", "auto"):
        MIN_NODES = 0
    try:
        yield
    finally:
        _MODE, MIN_NODES = old


def key_column(x) -> int | None:
    """Column of the fp32 node features ``x`` to order the nodes by, or None (policy off / batch too small /
    no such column)."""
    import torch

    if _MODE == "off" or x.dim() != 2 or x.dtype != torch.float32 or x.shape[0] < max(MIN_NODES, 2):
        return None
    col = AUTO_COLUMN if _MODE == "auto" else int(_MODE)
    return col if 0 <= col < x.shape[1] else None


def loaded_order(data):
    """Key column the events of ``data`` were renumbered by when they were loaded (``io.renumber_nodes``: one value
    per event after ``collate``), or None (not renumbered, or the events disagree)."""
    k = getattr(data, "node_order_key", None)
    if isinstance(k, (list, tuple)):
        return k[0] if k and all(v == k[0] for v in k) else None
    return k


def order_column(data) -> int | None:
    """THE predicate for "this batch is renumbered inside the step": the policy's column for ``data.x``, unless every
    event of the batch was already renumbered by that very column when it was read.  ``ECForGraphTCN`` and whoever
    builds the index ahead of it (``io.PrefetchLoader``, ``bench.py``) ask this one function, so a prefetched index
    is always the one the step looks up."""
    col = key_column(data.x)
    return None if col is None or loaded_order(data) == col else col
