"""SURVEY.md section 8(f) row 2 on the GPU: the ingestion kernels (csrc/ingest.hip) against golden g14 produced by the
reference's own process_tu_dataset / EdgeListData + _make_binorm_adj, and against the host restatements at sizes the
reference cannot ingest (its block-diagonal adjacency is dense)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def test_tu_batch_to_csr_matches_reference_g14(dev):
    from ragraph_amd.data import Data
    from ragraph_amd.ragraph_utils import process_tu_dataset

    g = gold("g14_ingestion")
    n = g["tu_x"].shape[0]
    batch = Data(torch.from_numpy(g["tu_x"]), torch.from_numpy(g["tu_edge_index"]))
    feats, adj, labels = process_tu_dataset(batch, int(g["tu_num_node_attributes"]), device=dev)
    dense = torch.zeros(n, n, device=dev)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), adj.rowptr[1:] - adj.rowptr[:-1])
    dense[rows, adj.col.long()] = adj.val
    assert np.allclose(dense.cpu().numpy(), g["tu_adj"], atol=1e-7)
    assert np.array_equal(dense.cpu().numpy() != 0, g["tu_adj"] != 0)
    assert np.array_equal(feats.cpu().numpy(), g["tu_features"]) and np.array_equal(labels.cpu().numpy(), g["tu_node_labels"])
    cols = adj.col.long()
    assert bool(((cols[1:] > cols[:-1]) | (rows[1:] != rows[:-1])).all())   # ascending columns within every row


def test_edge_tsv_to_edges_matches_reference_g14(dev, tmp_path):
    from ragraph_amd.edge_data import EdgeListData

    g = gold("g14_ingestion")
    tr, te = tmp_path / "train.txt", tmp_path / "test.txt"
    tr.write_text(str(g["edge_train_txt"]))
    te.write_text(str(g["edge_test_txt"]))
    ds = EdgeListData(str(tr), str(te), hour_interval=int(g["edge_hour_interval"]), device=dev)
    assert np.array_equal(ds.edges.cpu().numpy(), g["edge_edges"])
    assert np.array_equal(ds.edge_times.cpu().numpy(), g["edge_times"])
    assert np.allclose(ds.edge_norm.cpu().numpy(), g["edge_norm"], atol=1e-7)


def test_ingestion_kernels_match_host_restatement_at_scale(dev):
    """100k-node graph with duplicate edges (the config-2 graph) and 1M interactions with repeated (user, item) pairs:
    HIP == the torch / numpy restatements, bit for bit."""
    from ragraph_amd import kernels as K
    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.edge_data import binorm_edges
    from ragraph_amd.graph import CSRGraph

    n = 100_000
    ei = synthetic_big_graph(n, 10, seed=8, device=dev)
    ei = torch.cat([ei, ei[:, :5000]], dim=1)                     # duplicates: multiplicity 2 entries
    rowptr, col, val = K.csr_sym_normalized_from_edges(ei, n)
    ref = CSRGraph.from_edge_index_sym_normalized(ei.cpu(), n)    # CPU tensors -> the torch restatement
    assert torch.equal(rowptr.cpu(), ref.rowptr) and torch.equal(col.cpu(), ref.col)
    assert torch.equal(val.cpu(), ref.val)
    rng = np.random.default_rng(1)
    U, I, E = 30_000, 20_000, 1_000_000
    u = rng.integers(0, U, E)
    i = np.minimum((rng.pareto(1.1, E) * I / 50).astype(np.int64), I - 1)
    step = rng.integers(1, 700, E)
    e_ref, n_ref, t_ref = binorm_edges(U, I, u, i, step)
    e, nm, t = K.binorm_edges(torch.from_numpy(u).to(dev), torch.from_numpy(i).to(dev), torch.from_numpy(step).to(dev), U, I)
    assert np.array_equal(e.cpu().numpy(), e_ref) and np.array_equal(t.cpu().numpy(), t_ref)
    assert np.array_equal(nm.cpu().numpy(), n_ref)
