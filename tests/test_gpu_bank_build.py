"""SURVEY.md section 8(f) row 1 on the GPU: the toy-bank construction kernels (batched PageRank + degree inverse-importance
sampling probabilities, batched Floyd-Warshall position codes) against the C oracle bit for bit and against golden g13
from the reference; the batched build (node / graph flavours) and the edge flavour's vanilla-phase sampled bank."""
import os

import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _batch_of_graphs(rng, sizes):
    """Block-diagonal dense adjacency of len(sizes) random graphs: normalised-adjacency-like and 0/1 rewired ones with
    rows without out-edges."""
    n = sum(sizes)
    a = np.zeros((n, n), np.float32)
    off = 0
    for gi, s in enumerate(sizes):
        if gi % 2 == 0:
            b = (rng.random((s, s)) < 0.15).astype(np.float32)
            b = np.maximum(b, b.T) + np.eye(s, dtype=np.float32)
            d = b.sum(1) ** -0.5
            b = (b * d[:, None]) * d[None, :]
        else:
            b = (rng.random((s, s)) < 0.1).astype(np.float32)
            b[rng.integers(0, s)] = 0          # a dangling node
        a[off:off + s, off:off + s] = b
        off += s
    return a, np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def test_pagerank_sample_prob_batch_bit_exact_vs_oracle(dev):
    from ragraph_amd import kernels as K
    from ragraph_amd.bank_build import compute_sample_prob
    from ragraph_amd.graph import CSRGraph

    rng = np.random.default_rng(3)
    a, gp = _batch_of_graphs(rng, [41, 33, 7, 120, 2, 300, 64])
    g = CSRGraph.from_dense(T(a, dev))
    prob = compute_sample_prob(g, T(gp, dev))
    rowptr, col, val = cref.dense_to_csr(a)
    rt, ct, vt = cref.dense_to_csr_t(a)
    p_ref, it_ref = cref.pagerank(rt, ct, vt, cref.csr_row_sums(rowptr, val), gp)
    prob_ref = cref.sample_prob(p_ref, cref.csr_row_sums(rt, vt), gp)
    gt, _ = CSRGraph.from_coo(g.col.long(), torch.repeat_interleave(torch.arange(g.n, device=dev), g.rowptr[1:] - g.rowptr[:-1]),
                              g.val, g.n, sort_cols=True)
    p, iters = K.pagerank(gt.rowptr, gt.col, gt.val, K.csr_row_sums(g.rowptr, g.val), T(gp, dev))
    assert np.array_equal(iters.cpu().numpy(), it_ref) and int(iters.max()) < 128      # every graph converged
    assert np.array_equal(p.cpu().numpy(), p_ref)
    assert np.array_equal(prob.cpu().numpy(), prob_ref)
    for lo, hi in zip(gp[:-1], gp[1:]):                                                 # a distribution per graph
        assert abs(float(prob[lo:hi].sum()) - 1.0) < 1e-5


def test_bank_build_kernels_match_reference_g13(dev):
    from ragraph_amd import kernels as K
    from ragraph_amd.bank_build import compute_sample_prob
    from ragraph_amd.graph import CSRGraph

    g = gold("g13_bank_build")
    for tag in ("adj_norm", "adj_rewired"):
        prob = compute_sample_prob(CSRGraph.from_dense(T(g[tag], dev)))
        assert np.allclose(prob.cpu().numpy(), g[tag + "_sample_prob"], rtol=2e-5, atol=1e-7)
    n = 80
    dense = np.zeros((n, n), np.float32)
    dense[g["edge_adj_indices"][0], g["edge_adj_indices"][1]] = g["edge_adj_values"]
    prob = compute_sample_prob(CSRGraph.from_dense(T(dense, dev)))
    assert np.allclose(prob.cpu().numpy(), g["edge_sample_prob"], rtol=2e-5, atol=1e-7)
    codes, dist = K.position_codes_batch(T(g["sample_adj"][None], dev), T(g["anchors"][None], dev), 10.0, return_dist=True)
    assert np.allclose(codes[0].cpu().numpy(), g["position_codes"], atol=1e-6)
    assert np.array_equal(dist[0].cpu().numpy(), cref.floyd_warshall(g["sample_adj"]))
    # batched: 50 sampled toy graphs at once == the oracle graph by graph
    rng = np.random.default_rng(4)
    adj = (rng.random((50, 10, 10)) < 0.3).astype(np.float32) * rng.random((50, 10, 10)).astype(np.float32)
    anchors = rng.integers(0, 10, (50, 10))
    got = K.position_codes_batch(T(adj, dev), T(anchors, dev), 10.0)
    assert np.array_equal(got.cpu().numpy(), cref.position_codes_batch(adj, anchors))


def test_batched_build_toy_graph_node_and_graph(dev):
    from ragraph_amd import kernels as K
    from ragraph_amd.data import DataLoader, synthetic_tu_dataset
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.ragraph_utils import ToyGraphBase, process_tu_dataset, seed_everything

    seed_everything(1)
    ds = synthetic_tu_dataset(num_graphs=20, num_node_attributes=18, num_node_labels=3, num_classes=2, seed=5)
    pre = PrePrompt(18, 256, "prelu", 1, 0.3).to(dev)
    tgb = ToyGraphBase(pre, 3, 256, 3, device=dev, flavour="node")
    tgb.build_toy_graph(ds)
    G, S, V = 20, tgb.num_inverse_sample, 1 + tgb.num_augment_scale
    assert tgb.resource_keys.shape == (G * S * V, 256) and tgb.resource_positions.shape == (G * S * V, 10)
    assert tgb.resource_labels.shape == (G * S * V, 3) and torch.isfinite(tgb.resource_values).all()
    # the un-augmented variant comes first: each of its keys is the normalised embedding of a node of ITS graph
    batch = next(iter(DataLoader(ds, batch_size=20)))
    feats, adj, labels = process_tu_dataset(batch, 18, device=dev)
    emb = K.normalize_rows(pre.inference(feats, adj))
    ptr = batch.ptr.tolist()
    keys0 = tgb.resource_keys[:G * S].reshape(G, S, 256)
    for gi in range(G):
        cos = keys0[gi] @ emb[ptr[gi]:ptr[gi + 1]].t()
        assert bool((cos.max(dim=1).values > 1 - 1e-5).all())
    lab0 = tgb.resource_labels[:G * S]
    assert bool(((lab0 == 0) | (lab0 == 1)).all()) and bool((lab0.sum(1) == 1).all())
    pos = tgb.resource_positions
    assert bool(((pos >= 0) & (pos <= 1)).all()) and bool((pos[:G * S] > 0).any())
    # graph flavour: one (mean) row per resource graph, no sampling, no positions
    tg = ToyGraphBase(pre, 2, 256, 1, device=dev, flavour="graph")
    tg.build_toy_graph(ds)
    assert tg.resource_keys.shape == (20, 256) and tg.resource_labels.shape == (20, 2) and tg.resource_positions.shape[0] == 0
    keys_full = K.normalize_rows(pre.inference(feats, adj))
    want = K.segment_reduce(keys_full, batch.ptr.to(dev), mean_mode=True)
    assert torch.equal(tg.resource_keys, want)


def test_edge_vanilla_phase_sampled_bank(dev):
    from ragraph_amd.data import synthetic_bipartite
    from ragraph_amd.RAGraph_edge import RAGraph as RAGraphEdge

    U, I = 600, 400
    edges, norm, times = synthetic_bipartite(U, I, edges_per_user=6, seed=11, device=dev)

    class DS:
        num_users, num_items = U, I
    DS.edges, DS.edge_norm, DS.edge_times = edges, norm, times

    class Pre:
        def generate(self):
            g = torch.Generator(device=dev).manual_seed(3)
            return 0.1 * torch.randn(U, 64, device=dev, generator=g), 0.1 * torch.randn(I, 64, device=dev, generator=g)

    torch.manual_seed(0)
    S = round(0.01 * (U + I))                       # modules/RAGraph.py:44: num_inverse_sample = round(0.01 * len(adj))
    m = RAGraphEdge(DS, Pre(), phase="vanilla", use_RAG=True, retrieve_num=5, num_augment_scale=1, num_inverse_sample=S,
                    device=dev).eval()
    assert m.resource_keys.shape == (2 * S, 64) and m.resource_values.shape == (2 * S, 64)
    # sampling probabilities == the oracle's on the same bi-normalised adjacency
    prob = m.sample_prob().cpu().numpy()
    n = U + I
    dense = np.zeros((n, n), np.float32)
    e = edges.cpu().numpy()
    dense[e[:, 0], e[:, 1]] = norm.cpu().numpy()
    ref, _, it = cref.compute_sample_prob_dense(dense)
    assert np.array_equal(prob, ref) and int(it[0]) < 128
    # the un-augmented half of the bank consists of rows of the propagated embeddings
    full = RAGraphEdge(DS, Pre(), phase="vanilla", use_RAG=True, retrieve_num=5, device=dev)
    same = (m.resource_keys[:S].unsqueeze(1) == full.resource_keys.unsqueeze(0)).all(dim=-1)
    assert bool(same.any(dim=1).all())
    uo, io = m.generate()
    assert torch.isfinite(uo).all() and torch.isfinite(io).all() and uo.shape == (U, 64)



def test_sample_prob_raises_when_pagerank_does_not_converge(dev, monkeypatch):
    """The reference iterates PageRank until convergence (InverseSampling.py:38-44); the batched form enqueues a fixed
    number of iterations and must not hand back an unconverged iterate silently."""
    from ragraph_amd import bank_build
    from ragraph_amd.graph import CSRGraph

    rng = np.random.default_rng(3)
    a = (rng.random((60, 60)) < 0.1).astype(np.float32)
    a = np.maximum(a, a.T)
    g = CSRGraph.from_dense(T(a, dev))
    p = bank_build.compute_sample_prob(g)                       # converges well inside the default budget
    assert torch.isfinite(p).all() and abs(float(p.sum()) - 1.0) < 1e-5
    monkeypatch.setattr(bank_build, "PAGERANK_MAX_ITER", 3)
    with pytest.raises(RuntimeError, match="did not converge"):
        bank_build.compute_sample_prob(g)
    assert torch.isfinite(bank_build.compute_sample_prob(g, check=False)).all()
