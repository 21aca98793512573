"""Per-edge model outputs that are HELD in the graph index's CSR (target-sorted) order and
PRESENT themselves in ``edge_index`` order.

The reference returns ``W`` / ``edge_embedding`` in the order of ``data.edge_index``
(models/edge_classifier.py:108-121).  Inside this package every edge tensor lives in CSR order,
and scattering 64 M four-byte weights back into ``edge_index`` order (and gathering their
gradient again) costs three times the head kernels' algorithmic traffic - for a tensor whose
only consumer in training is a mean over all edges.  ``EdgeOrdered`` therefore defers the
scatter:

* the losses of this package recognise it and run on the CSR-ordered values directly (labels are
  gathered into CSR order once per batch, ``ops.edge_targets_csr``);
* ANY other use - arithmetic, indexing, ``.cpu()``, printing, comparison, ``torch.*`` functions,
  a third-party loss - goes through ``__torch_function__``, which replaces it by the real
  ``edge_index``-ordered tensor (one row scatter, differentiable, done at most once).

Shape, dtype and device queries are answered from the metadata without materialising.
"""

from __future__ import annotations

import torch
from torch import Tensor
from torch.utils._pytree import tree_map

# attribute / method names that only look at metadata
_META = frozenset({"shape", "dtype", "device", "ndim", "is_cuda", "layout", "requires_grad",
                   "size", "dim", "numel", "nelement", "__len__", "element_size", "is_floating_point",
                   "is_complex", "is_sparse", "is_quantized", "is_meta", "names", "itemsize", "nbytes",
                   "is_contiguous", "stride", "storage_offset", "is_cpu", "is_nested"})
# (``is_leaf``, ``grad``, ``grad_fn``, ``retain_grad``, ``_version`` are NOT metadata of the wrapper: they
#  are answered by the materialised tensor, which is the one autograd knows)


def _fname(func) -> str:
    n = getattr(func, "__name__", "")
    if n in ("__get__", "__set__"):
        n = getattr(getattr(func, "__self__", None), "__name__", n)
    return n


class EdgeOrdered(Tensor):
    @staticmethod
    def __new__(cls, csr: Tensor, gi):
        r = Tensor._make_wrapper_subclass(cls, csr.shape, dtype=csr.dtype, device=csr.device,
                                          requires_grad=False)
        r._csr, r._gi, r._coo = csr, gi, None
        return r

    #: the values in CSR order (autograd-tracked) and the graph index they are ordered by
    @property
    def csr(self) -> Tensor:
        return self._csr

    @property
    def graph_index(self):
        return self._gi

    def in_edge_index_order(self) -> Tensor:
        """The ordinary tensor in ``edge_index`` order (row scatter through the CSR permutation)."""
        if self._coo is None:
            from . import ops

            if self._gi.n_edges <= 1:   # nothing to reorder (a lone weight squeezes to 0-dim)
                self._coo = self._csr
            else:
                with torch._C.DisableTorchFunctionSubclass():
                    self._coo = ops.permute_rows(self._csr, self._gi.perm, scatter=True)
        return self._coo

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = _fname(func)
        if name in _META:
            with torch._C.DisableTorchFunctionSubclass():
                if name == "requires_grad" and getattr(func, "__name__", "") == "__get__":
                    return args[0]._csr.requires_grad
                return func(*args, **kwargs)

        def real(a):
            return a.in_edge_index_order() if isinstance(a, EdgeOrdered) else a

        with torch._C.DisableTorchFunctionSubclass():
            return func(*tree_map(real, args), **tree_map(real, kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # backstop only: every Python-level use is caught by __torch_function__ above
        def real(a):
            return a.in_edge_index_order() if isinstance(a, EdgeOrdered) else a

        return func(*tree_map(real, args), **tree_map(real, kwargs or {}))

    def __repr__(self):  # (tensor printing walks the storage)
        return f"EdgeOrdered({self.in_edge_index_order()!r})"

    __str__ = __repr__

    def __reduce_ex__(self, proto):
        return self.in_edge_index_order().__reduce_ex__(proto)


class NodeOrdered(EdgeOrdered):
    """The per-node twin (round 5): a node tensor held in the RENUMBERED node order of a graph index built with
    ``order_by`` (locality.py) that presents itself in the caller's node order - ``node_embedding`` of the edge
    classifier, which nothing reads in edge-classifier training; any use gathers it through ``node_rank`` once."""

    def in_edge_index_order(self) -> Tensor:   # (here: the caller's NODE order)
        if self._coo is None:
            from . import ops

            with torch._C.DisableTorchFunctionSubclass():
                self._coo = ops.permute_rows(self._csr, self._gi.node_rank)
        return self._coo

    def __repr__(self):
        return f"NodeOrdered({self.in_edge_index_order()!r})"

    __str__ = __repr__


def as_tensor(t):
    """``t`` itself, or the ``edge_index``-ordered tensor behind an ``EdgeOrdered``."""
    return t.in_edge_index_order() if isinstance(t, EdgeOrdered) else t
