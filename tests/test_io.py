"""Reference graph files (.pt, PyG-pickled Data) without PyG: restricted unpickler,
dataset listing, prefetch loader.  The fixture
tests/golden/test_graph.pt is the reference's own test graph
(tests/test_data/graphs/test_graph.pt)."""

import os
import pathlib
import pickle
import shutil

import numpy as np
import pytest
import torch

from gnn_tracking_amd import io as gio

GOLD = pathlib.Path(__file__).resolve().parent / "golden"


def test_load_reference_graph_matches_golden():
    d = gio.load_graph(GOLD / "test_graph.pt")
    z = np.load(GOLD / "g1_ec_testgraph.npz")
    for k in ("x", "edge_index", "edge_attr", "y", "pt"):
        assert torch.equal(getattr(d, k), torch.from_numpy(z[k])), k
    assert d.num_nodes == 90 and d.num_edges == 253
    for k in ("particle_id", "reconstructable", "sector", "eta", "layer"):
        assert k in d


def test_malicious_pickle_is_refused(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > " + str(tmp_path / "pwned"),))

    p = tmp_path / "evil.pt"
    torch.save({"x": torch.zeros(1), "payload": Evil()}, p)
    with pytest.raises(pickle.UnpicklingError):
        gio.load_graph(p)
    assert not (tmp_path / "pwned").exists()


def test_dataset_and_prefetch_loader(tmp_path):
    for i in range(5):
        shutil.copy(GOLD / "test_graph.pt", tmp_path / f"data2100{i}_s{i % 2}.pt")
    ds = gio.GraphDataset(tmp_path)
    assert len(ds) == 5 and len(gio.GraphDataset(tmp_path, sector=1)) == 2
    assert len(gio.GraphDataset(tmp_path, start=1, stop=3)) == 2
    batches = list(gio.PrefetchLoader(ds, batch_size=2, depth=2))
    assert [b.num_nodes for b in batches] == [180, 180, 90]
    assert batches[0].edge_index.max().item() >= 90  # second graph offset by collate
    assert torch.equal(batches[2].x, ds[4].x)


@pytest.mark.gpu
def test_prefetch_loader_to_device(tmp_path):
    for i in range(4):
        shutil.copy(GOLD / "test_graph.pt", tmp_path / f"g{i}.pt")
    ds = gio.GraphDataset(tmp_path)
    ref = ds[0]
    n = 0
    for b in gio.PrefetchLoader(ds, batch_size=1, device="cuda:0", depth=2):
        assert b.x.is_cuda and torch.equal(b.x.cpu(), ref.x) and torch.equal(b.edge_index.cpu(), ref.edge_index)
        n += 1
    assert n == 4
    # the loader can also build the graph index of a staged batch on its side stream
    from gnn_tracking_amd import ops
    for b in gio.PrefetchLoader(ds, batch_size=2, device="cuda:0", depth=2, build_index=True):
        hit = ops._GI_CACHE.get((id(b.edge_index), False))   # (key: the tensor, renumbered build or not)
        assert hit is not None and hit[3].ready is not None, "index was not prefetched"
        gi = ops.graph_index(b.edge_index, b.num_nodes)
        assert gi is hit[3] and gi.ready is None
        ref_gi = ops.graph_index(b.edge_index, b.num_nodes, cache=False)
        for k in ("perm", "tgt", "src", "rowptr_t", "rowptr_s", "spos", "spos_inv"):
            assert torch.equal(getattr(gi, k), getattr(ref_gi, k)), k

