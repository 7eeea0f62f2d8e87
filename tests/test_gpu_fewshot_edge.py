"""Section 8(f) widening rows 3-4 on the GPU: top-k over a materialised matrix, history masking, Floyd-Warshall /
position codes, the few-shot retrieve (golden G8 from the reference) and the edge recommendation evaluation."""
import os

import numpy as np
import pytest
import torch

from oracle import cref, pipeline

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.mark.parametrize("B,N,k", [(1, 10, 10), (5, 1000, 20), (64, 107029, 20), (7, 4099, 64), (3, 257, 1)])
def test_topk_rows_bit_exact(dev, B, N, k):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(B + N + k)
    S = rng.standard_normal((B, N)).astype(np.float32)
    S[0, : min(N, 300)] = 0.5  # a run of exact ties
    s, i = K.topk_rows(T(S, dev), k)
    rs, ri = cref.topk_rows(S, k)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    # a row slice (unaligned rows) goes through the scalar path
    if N > 8:
        s2, i2 = K.topk_rows(T(S, dev)[:, 1:N - 2].contiguous(), min(k, N - 3))
        r2s, r2i = cref.topk_rows(S[:, 1:N - 2], min(k, N - 3))
        assert np.array_equal(i2.cpu().numpy(), r2i) and np.array_equal(s2.cpu().numpy(), r2s)


def test_topk_rows_agrees_with_fused_kernel(dev):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(0)
    kn = cref.normalize_rows(rng.standard_normal((3000, 128)).astype(np.float32))
    q = rng.standard_normal((50, 128)).astype(np.float32)
    knd = T(kn, dev)
    s_f, i_f = K.topk_cosine(T(q, dev), knd, 10)
    scores = K.linear(K.normalize_rows(T(q, dev)), knd)   # same fmaf chains, materialised
    s_r, i_r = K.topk_rows(scores, 10)
    assert torch.equal(i_f, i_r) and torch.equal(s_f, s_r)


def test_floyd_warshall_and_position_code(dev):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(5)
    n = 97
    a = (rng.random((n, n)) < 0.05).astype(np.float32) * rng.random((n, n)).astype(np.float32)
    a = np.maximum(a, a.T)
    np.fill_diagonal(a, 0.3)
    d = K.floyd_warshall(T(a, dev))
    rd = cref.floyd_warshall(a)
    assert np.array_equal(d.cpu().numpy(), rd)
    anchors = rng.integers(0, n, 10)
    pc = K.position_code(d, T(anchors, dev), 10.0)
    assert np.array_equal(pc.cpu().numpy(), cref.position_code(rd, anchors, 10.0))


@pytest.mark.parametrize("n,deg", [(97, 0.05), (528, 0.01), (3000, 0.002)])
def test_position_codes_csr_bit_exact_vs_oracle(dev, n, deg):
    """Distances to the anchors only (the few-shot flavour's per-forward path): HIP == oracle bit for bit on graphs with
    unreachable nodes, explicit zero entries and duplicate anchors; within 1e-6 of the all-pairs route."""
    from ragraph_amd import kernels as K
    from ragraph_amd.graph import CSRGraph

    rng = np.random.default_rng(n)
    a = (rng.random((n, n)) < deg).astype(np.float32) * rng.random((n, n)).astype(np.float32)
    a = np.maximum(a, a.T)
    np.fill_diagonal(a, 0.3)
    a[n // 2:, : n // 2] = 0                      # two halves: one direction cut, nodes that cannot reach some anchors
    rowptr, col, val = cref.dense_to_csr(a)
    val = val.copy()
    val[::17] = 0.0                               # explicit zeros in the CSR are "no edge" (dist[adj == 0] = inf)
    anchors = rng.integers(0, n, 10)
    anchors[3] = anchors[2]
    oc, od = cref.position_codes_csr(rowptr, col, val, anchors, 10.0)
    codes, dist = K.position_codes_csr(T(rowptr, dev), T(col, dev), T(val, dev), T(anchors, dev), 10.0, return_dist=True)
    assert np.array_equal(dist.cpu().numpy(), od) and np.array_equal(codes.cpu().numpy(), oc)
    assert np.isinf(od).any() and np.isfinite(od).any()
    if n <= 600:                                  # the reference's all-pairs route on the same (zero-cleaned) matrix
        dense = np.zeros((n, n), dtype=np.float32)
        rows = np.repeat(np.arange(n), np.diff(rowptr))
        dense[rows, col] = val
        fw = cref.floyd_warshall(dense)[:, anchors]
        assert np.array_equal(np.isinf(fw), np.isinf(od))
        assert np.allclose(od[np.isfinite(fw)], fw[np.isfinite(fw)], rtol=1e-6, atol=0)
        assert np.allclose(oc, cref.position_code(cref.floyd_warshall(dense), anchors, 10.0), atol=1e-6)
    g = CSRGraph(T(rowptr, dev), T(col, dev).to(torch.int32), T(val, dev), n)
    from ragraph_amd.RAGraph_fewshot import PositionAwareEncoder
    assert torch.equal(PositionAwareEncoder.encode_position_aware_code(g, 10, 10, anchors=T(anchors, dev)), codes)


def test_fewshot_retrieve_g8(dev):
    from ragraph_amd.RAGraph_fewshot import PositionAwareEncoder, ToyGraphBaseFewShot

    g = dict(np.load(os.path.join(GOLD, "g8_fewshot_retrieve.npz")))
    tgb = ToyGraphBaseFewShot(None, g["labels"].shape[1], 256, 3, int(g["k"]), device=dev)
    tgb.add_resources(T(g["keys"], dev), T(g["values"], dev), T(g["labels"], dev), T(g["positions"], dev))
    adj = T(g["adj"], dev)
    assert np.array_equal(PositionAwareEncoder.floyd_warshall(adj).cpu().numpy(), g["dist"])
    pos = PositionAwareEncoder.encode_position_aware_code(adj, 10, 10, anchors=T(g["anchors"], dev))
    assert np.allclose(pos.cpu().numpy(), g["pos_codes"], atol=1e-6)    # anchors-only distances: ~1 ulp from the all-pairs sums
    e, l = tgb.retrieve(T(g["Q"], dev), adj, False, anchors=T(g["anchors"], dev))
    assert np.array_equal(e.cpu().numpy(), g["rag_embeddings"]) and np.array_equal(l.cpu().numpy(), g["rag_labels"])
    # bit-exact vs the oracle's restatement of the mixed score
    sc = tgb.similarity_scores(T(g["Q"], dev), adj, anchors=T(g["anchors"], dev))
    osc, _ = pipeline.fewshot_scores(g["Q"], g["adj"], g["anchors"], g["keys"], g["positions"])
    assert np.array_equal(sc.cpu().numpy(), osc)


def test_fewshot_forward_runs_and_matches_oracle_composition(dev):
    from ragraph_amd import kernels as K
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph_fewshot import RAGraph as RAGraphFewShot

    g = dict(np.load(os.path.join(GOLD, "g8_fewshot_retrieve.npz")))
    torch.manual_seed(0)
    F_in, C, D = 18, 3, 256
    pre = PrePrompt(F_in, D, "prelu", 2, 0.3).to(dev)     # few-shot uses two layers: encode = 0, decode = 1
    mean_logits = torch.randn(C, D, device=dev)
    model = RAGraphFewShot(pre, None, mean_logits, D, device=dev, dataset_name="ENZYMES").eval()
    model.toy_graph_base.add_resources(T(g["keys"], dev), T(g["values"], dev), T(g["labels"], dev), T(g["positions"], dev))
    X = torch.rand(g["adj"].shape[0], F_in, device=dev)
    adj, anchors = T(g["adj"], dev), T(g["anchors"], dev)
    with torch.no_grad():
        out = model(X, adj, mean_logits, anchors=anchors)
        emb = pre.encode(X, adj)
    assert out.shape == (X.shape[0], D) and torch.isfinite(out).all()
    # oracle composition of RAGraph_node_fewshot/RAGraph.py:47-79
    c0, c1 = pre.gcn.convs
    csr = cref.dense_to_csr(g["adj"])
    oe = pipeline.gcn_layer(X.cpu().numpy(), csr, c0.fc.weight.detach().cpu().numpy(), c0.bias.detach().cpu().numpy(),
                            float(c0.act.weight.detach()))
    assert np.array_equal(emb.cpu().numpy(), oe)
    re_, rl, idx, _ = pipeline.fewshot_retrieve(oe, g["adj"], g["anchors"], g["keys"], g["values"], g["labels"],
                                                g["positions"], 5)
    rag_logits, _ = cref.gather_reduce(mean_logits.cpu().numpy(), None, rl.argmax(-1), v_scale=np.float32(1 / 5))
    rag_emb = re_.reshape(-1, 5, D)
    acc = np.zeros((rag_emb.shape[0], D), dtype=np.float32)
    for j in range(5):
        acc = acc + rag_emb[:, j]
    hidden = cref.axpby(pipeline.propagate(csr, oe, 3), 0.5, acc, 0.5)
    dec = pipeline.gcn_layer(hidden, csr, c1.fc.weight.detach().cpu().numpy(), c1.bias.detach().cpu().numpy(),
                             float(c1.act.weight.detach()))
    ref = cref.axpby(dec, 0.5, rag_logits, 0.5)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_edge_eval_topk_items(dev):
    from ragraph_amd import edge_eval

    rng = np.random.default_rng(1)
    U, I, D, k = 700, 3001, 64, 20
    ue = rng.standard_normal((U, D)).astype(np.float32)
    ie = rng.standard_normal((I, D)).astype(np.float32)
    users = rng.permutation(U)[:600]
    hist = [rng.choice(I, size=rng.integers(0, 30), replace=False) for _ in users]
    rowptr = np.concatenate([[0], np.cumsum([len(h) for h in hist])]).astype(np.int64)
    cols = np.concatenate(hist).astype(np.int64) if len(hist) else np.zeros(0, np.int64)
    idx = edge_eval.topk_items(None, T(users, dev), T(rowptr, dev), T(cols, dev), k=k, eval_batch_size=256,
                               embeddings=(T(ue, dev), T(ie, dev)))
    ref = pipeline.edge_topk_items(ue, ie, users, hist, k)
    assert np.array_equal(idx.cpu().numpy(), ref)
    for r, h in enumerate(hist):   # no history item is ever recommended
        assert not np.isin(idx[r].cpu().numpy(), h).any()
    truth = [set(rng.choice(I, 5, replace=False).tolist()) for _ in users]
    rec, ndcg = edge_eval.recall_ndcg(idx.cpu().numpy(), truth, k)
    assert 0.0 <= rec <= 1.0 and 0.0 <= ndcg <= 1.0


def test_topk_select_rows_matches_oracle(dev):
    """The canonical top-k SET for large k: ties at the k-th place (duplicates, constant rows), k = 1, k = N, negative and
    zero scores, row strides that are not multiples of 256 -- index sets bit-exact vs the oracle."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(12)
    for (B, N, k) in [(5, 1000, 1), (7, 1000, 1000), (9, 5000, 1000), (3, 70001, 33333), (4, 513, 257), (2, 100, 64), (3, 300001, 100000), (2, 65536, 1), (2, 131072, 131072)]:
        S = rng.standard_normal((B, N)).astype(np.float32)
        S[0, : N // 2] = S[0, N // 2: 2 * (N // 2)]        # exact duplicates: ties everywhere
        S[1] = 0.25                                          # a constant row: the first k indices win
        if B > 2:
            S[2, ::3] = 0.0
            S[2, 1::3] = -0.0
        kth, idx = K.topk_select_rows(torch.from_numpy(S).to(dev), k)
        rk, ri = cref.topk_select_rows(S, k)
        assert np.array_equal(idx.cpu().numpy(), ri), (B, N, k)
        assert np.array_equal(kth.cpu().numpy(), rk), (B, N, k)


def test_retrieve_mean_very_large_k_blocked_sum(dev):
    """retrieve_num beyond ROW_BLOCK (the reference's amazon setting is 100000): the winners' sum takes the SpMM hub-row
    path -- blocks of 4096 winners in ascending index order, block sums added in order.  Bit-identical to the oracle's
    blocked SpMM over the oracle's canonical top-k set, and the plain mean to 1e-5."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(31)
    B, N, D, k = 6, 30000, 64, 10000
    kn = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    v = rng.standard_normal((N, D), dtype=np.float32)
    q = rng.standard_normal((B, D), dtype=np.float32)
    got = K.retrieve_mean_large_k(T(q, dev), T(kn, dev), T(v, dev), k).cpu().numpy()
    S = cref.linear(cref.normalize_rows(q), kn)
    _, idx = cref.topk_select_rows(S, k)
    rowptr = np.arange(0, (B + 1) * k, k, dtype=np.int64)
    total = cref.spmm_csr(rowptr, idx.reshape(-1).astype(np.int32), np.ones(B * k, np.float32), v)
    want = cref.axpby(total, 1.0 / k, total, 0.0)
    assert np.array_equal(got, want)
    assert np.allclose(got, np.stack([v[idx[b]].astype(np.float64).mean(0) for b in range(B)]), atol=1e-5)


def test_edge_vanilla_large_k_g12(dev):
    """RAGraph_edge vanilla phase, retrieve_num = 1000 over a 5000-row bank: the reference's generate() (golden g12) and
    the oracle composition (retrieval part bit for bit: test_topk_select_rows_matches_oracle)."""
    from ragraph_amd.RAGraph_edge import RAGraph as RAGraphEdge

    g = gold("g12_edge_large_k")
    U, I = int(g["num_users"]), int(g["num_items"])

    class DS:
        num_users, num_items = U, I
        edges, edge_norm, edge_times = T(g["edges"], dev), T(g["edge_norm"], dev), T(g["edge_times"], dev)

    class Pre:
        def generate(self):
            return T(g["user_embedding"], dev), T(g["item_embedding"], dev)

    m = RAGraphEdge(DS, Pre(), phase="vanilla", use_RAG=False, retrieve_num=int(g["retrieve_num"]),
                    retrieve_weight=float(g["retrieve_weight"]), device=dev).eval()
    m.use_RAG = True
    m.resource_keys, m.resource_values = T(g["resource_keys"], dev), T(g["resource_values"], dev)
    uo, io = m.generate()
    out = torch.cat([uo, io]).cpu().numpy()
    ref = np.concatenate([g["user_out"], g["item_out"]])
    ok = g["boundary_gap"] > 1e-6
    assert np.allclose(out[ok], ref[ok], atol=2e-5)
    all_emb = np.concatenate([g["user_embedding"], g["item_embedding"]])
    o, _, _, _, _ = pipeline.edge_forward(g["edges"], g["edge_norm"], g["edge_times"], all_emb, g["resource_keys"],
                                          g["resource_values"], int(g["retrieve_num"]), float(g["retrieve_weight"]), 3)
    assert np.allclose(out, o, atol=1e-6)   # (the time softmax calls expf: oracle and device libm agree to 1e-6)
