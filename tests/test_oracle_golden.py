"""Pins the CPU oracle (oracle/ragraph_oracle.c + oracle/pipeline.py + oracle/ref_torch.py) against golden vectors
produced by the reference itself (oracle/make_golden.py, run in the build container).  No GPU.

Bar: top-k indices identical (fixtures are tie-free: adjacent top-(k+1) gaps > 1e-5); floats within 1e-5 absolute
(the reference's own summation order is MKL's; the oracle's is the fixed fmaf chain of include/ragraph_hip.h).
"""
import os

import numpy as np
import pytest
import torch

from oracle import cref, pipeline, ref_torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


@pytest.mark.parametrize("name", ["g1a_cosine_topk", "g1b_cosine_topk", "g1c_cosine_topk"])
def test_g1_cosine_topk(name):
    g = gold(name)
    kn = cref.normalize_rows(g["K"])
    full = cref.cosine_scores(cref.normalize_rows(g["Q"][:8]), kn)
    assert np.allclose(full, g["scores_full_first8"], atol=2e-6)
    for k in (1, 4, 5, 10):
        s, i = cref.topk_cosine(g["Q"], kn, k)
        assert np.array_equal(i, g[f"topk_idx_k{k}"])
        assert np.allclose(s, g[f"topk_scores_k{k}"], atol=2e-6)
    # the torch restatement is the reference's op chain: same indices, scores to the last bits
    e, _, i = ref_torch.retrieve(torch.from_numpy(g["Q"]), torch.from_numpy(g["K"]), torch.from_numpy(g["K"]), None, 10,
                                 slab=17)
    assert np.array_equal(i.numpy(), g["topk_idx_k10"])


def test_g3_duplicate_keys_values_only():
    g = gold("g3_duplicate_keys")
    e, l, idx = pipeline.retrieve(g["Q"], g["keys"], g["values"], g["labels"], int(g["k"]))
    assert np.allclose(e.sum(1), g["sum_values"], atol=1e-5)
    assert np.allclose(l.mean(1), g["mean_labels"], atol=1e-6)
    # canonical tie rule: among equal scores the lower index comes first
    kn = cref.normalize_rows(g["keys"])
    s, i = cref.topk_cosine(g["Q"], kn, int(g["k"]))
    same = s[:, :-1] == s[:, 1:]
    assert same.any(), "fixture is expected to contain exact ties"
    assert (i[:, :-1][same] < i[:, 1:][same]).all()


def test_g4_gcn_layer():
    g = gold("g4_gcn_layer")
    H = pipeline.gcn_layer(g["X"], cref.dense_to_csr(g["adj"]), g["W"], g["bias"], g["alpha"][0])
    assert np.allclose(H, g["H"], atol=1e-5)
    Ht = ref_torch.gcn_layer(*(torch.from_numpy(g[k]) for k in ("X", "adj", "W", "bias", "alpha")))
    assert np.allclose(Ht.numpy(), g["H"], atol=1e-6)


def test_g5_propagation():
    g = gold("g5_propagation")
    csr = cref.dense_to_csr(g["adj"])
    for k in (0, 1, 2, 3):
        assert np.allclose(pipeline.propagate(csr, g["x"], k), g[f"y_k{k}"], atol=1e-5)
        assert np.allclose(ref_torch.propagate(torch.from_numpy(g["adj"]), torch.from_numpy(g["x"]), k).numpy(),
                           g[f"y_k{k}"], atol=1e-6)


def test_g6_node_forward():
    g = gold("g6_node_forward")
    csr = cref.dense_to_csr(g["adj"])
    p = {k: g[k] for k in ("W", "bias", "fc1_w", "fc1_b", "fc2_w", "fc2_b")}
    p["alpha"] = g["alpha"][0]
    assert np.allclose(pipeline.decoder(g["dec_in"], g["fc1_w"], g["fc1_b"], g["fc2_w"], g["fc2_b"]), g["dec_out"],
                       atol=1e-5)
    logits, idx, h = pipeline.node_forward(g["X"], csr, p, g["keys"], g["values"], g["labels"], int(g["k"]),
                                           int(g["hops"]), float(g["retrieve_weight"]), float(g["label_weight"]))
    assert np.array_equal(idx, g["topk_idx"])
    assert np.allclose(logits, g["logits"], atol=1e-5)
    e, l, _ = pipeline.retrieve(h, g["keys"], g["values"], g["labels"], int(g["k"]))
    assert np.array_equal(e, g["rag_embeddings"]) and np.array_equal(l, g["rag_labels"])
    tp = {k: torch.from_numpy(np.asarray(v)) for k, v in p.items()}
    tl, ti = ref_torch.node_forward(torch.from_numpy(g["X"]), torch.from_numpy(g["adj"]), tp, torch.from_numpy(g["keys"]),
                                    torch.from_numpy(g["values"]), torch.from_numpy(g["labels"]), int(g["k"]),
                                    int(g["hops"]), float(g["retrieve_weight"]), float(g["label_weight"]), slab=64)
    assert np.array_equal(ti.numpy(), g["topk_idx"]) and np.allclose(tl.numpy(), g["logits"], atol=1e-6)


def test_g7_graph_forward():
    g = gold("g7_graph_forward")
    csr = cref.dense_to_csr(g["adj"])
    p = {k: g[k] for k in ("W", "bias", "fc1_w", "fc1_b", "fc2_w", "fc2_b")}
    p["alpha"] = g["alpha"][0]
    logits, idx, h = pipeline.graph_forward(g["X"], csr, p, g["keys"], g["values"], g["labels"], int(g["k"]), 1,
                                            float(g["retrieve_weight"]), float(g["label_weight"]))
    assert np.allclose(h, g["H"], atol=1e-5)
    assert np.array_equal(idx.reshape(-1), g["topk_idx"].reshape(-1))
    assert np.allclose(logits, g["logits"], atol=1e-5)
    e, l, _ = pipeline.retrieve(h.mean(0), g["keys"], g["values"], g["labels"], int(g["k"]))
    assert e.shape == g["rag_embeddings"].shape  # (1, k, D): 1-D query -> unsqueezed scores


def test_g9_edge_generate():
    g = gold("g9_edge_generate")
    out, idx, layers, tn, (rowptr, col, perm) = pipeline.edge_forward(
        g["edges"], g["edge_norm"], g["edge_times"], g["gated_emb"], g["resource_keys"], g["resource_values"], 10,
        float(g["retrieve_weight"]), int(g["num_layers"]))
    assert np.allclose(tn, g["time_norm"][perm], atol=1e-6)
    assert np.allclose(layers[1], g["agg1"], atol=1e-6)
    ok = g["row_gap"] > 1e-5  # rows whose reference top-11 has no near-tie (98.6 % of them)
    assert ok.mean() > 0.9
    assert np.array_equal(idx[ok], g["topk_idx"][ok])
    ref = np.concatenate([g["user_out"], g["item_out"]], 0)
    assert np.allclose(out[ok], ref[ok], atol=1e-5)
    assert np.allclose(out, ref, atol=5e-3)  # near-tie rows differ only by swapping two almost-equal neighbours


def test_g10_downprompt():
    g = gold("g10_downprompt")
    for C in (2, 6):
        logp, emb = pipeline.downprompt_logits(g["h"], g["w"], g["graph_len"], g[f"proto_c{C}"])
        assert np.allclose(emb, g["graph_emb"], atol=1e-4)
        assert np.allclose(logp, g[f"logp_c{C}"], atol=1e-5)


def test_g16_downprompt_node():
    g = gold("g16_downprompt_node")
    ave = pipeline.downprompt_node_averageemb(g["labels"], g["feature"])
    assert np.allclose(ave, g["ave_init"], rtol=1e-5, atol=1e-6)
    probs, rawret = pipeline.downprompt_node_forward(g["h"], g["w"], g["ave_init"])
    assert np.allclose(rawret, g["elu_wh"], atol=1e-6)
    assert np.allclose(probs, g["probs"], atol=1e-6)
    assert np.allclose(pipeline.downprompt_node_forward(g["h"], g["w"], g["ave_injected"])[0], g["probs_injected"], atol=1e-6)
    ave_t = pipeline.downprompt_node_averageemb(g["labels"], rawret)           # train=1: prototypes from this batch
    assert np.allclose(ave_t, g["ave_train"], rtol=1e-5, atol=1e-6)
    assert np.allclose(cref.proto_cosine(rawret, ave_t, mode=1), g["probs_train"], atol=1e-6)


def test_g8_fewshot_structural_retrieve():
    g = gold("g8_fewshot_retrieve")
    fw = cref.floyd_warshall(g["adj"])
    assert np.array_equal(fw, g["dist"])                                     # min-plus closure: exact
    # the reference's route (all-pairs matrix) and the product's per-forward route (distances to the anchors only):
    # the same paths, sums associated differently -- both within 1e-6 of the reference's codes, identical indices
    pos_fw = cref.position_code(fw, g["anchors"], 10.0)
    pos_a, dist_a = cref.position_codes_csr(*cref.dense_to_csr(g["adj"]), g["anchors"], 10.0)
    assert np.allclose(pos_fw, g["pos_codes"], atol=1e-7) and np.allclose(pos_a, g["pos_codes"], atol=1e-6)
    ref_d = g["dist"][:, g["anchors"]]
    assert np.array_equal(np.isinf(dist_a), np.isinf(ref_d))
    assert np.allclose(dist_a[np.isfinite(ref_d)], ref_d[np.isfinite(ref_d)], rtol=1e-6, atol=0)
    for all_pairs in (False, True):
        sc, pos = pipeline.fewshot_scores(g["Q"], g["adj"], g["anchors"], g["keys"], g["positions"], all_pairs=all_pairs)
        assert np.array_equal(cref.topk_rows(sc, int(g["k"]))[1], g["topk_idx"])
    e, l, idx, pos = pipeline.fewshot_retrieve(g["Q"], g["adj"], g["anchors"], g["keys"], g["values"], g["labels"],
                                               g["positions"], int(g["k"]))
    assert np.allclose(pos, g["pos_codes"], atol=1e-6)
    assert np.array_equal(idx, g["topk_idx"])
    assert np.array_equal(e, g["rag_embeddings"]) and np.array_equal(l, g["rag_labels"])


def test_topk_rows_matches_torch_topk_on_tie_free_rows():
    rng = np.random.default_rng(3)
    S = rng.standard_normal((33, 5000)).astype(np.float32)
    s, i = cref.topk_rows(S, 20)
    ts, ti = torch.topk(torch.from_numpy(S), 20)
    assert np.array_equal(i, ti.numpy()) and np.array_equal(s, ts.numpy())
    S[:, ::7] = 0.25  # ties: canonical order = lower index first
    s, i = cref.topk_rows(np.minimum(S, 0.25), 10)
    assert (s == 0.25).all() and np.array_equal(i[0], np.sort(i[0]))


def test_g11_add_noise_branches():
    """retrieve(add_noise=True): the deterministic prefix is the oracle's top-k' and the noise comes from torch's
    default CPU generator (ToyGraphBase.py:66,73-79; graph flavour :84-85,131-134)."""
    g = gold("g11a_node_noise")
    k, nz = int(g["retrieve_num"]), int(g["noise_retrieve_num"])
    e, l, idx = pipeline.retrieve(g["H"], g["keys"], g["values"], g["labels"], 2 * k)
    assert np.array_equal(e, g["rag_embeddings"][:, :2 * k]) and np.array_equal(l, g["rag_labels"][:, :2 * k])
    torch.manual_seed(int(g["seed"]))
    noise = torch.randint(0, g["values"].shape[0], (g["H"].shape[0], nz)).numpy()
    assert np.array_equal(g["values"][noise], g["rag_embeddings"][:, 2 * k:])
    assert np.array_equal(g["labels"][noise], g["rag_labels"][:, 2 * k:])

    g = gold("g11b_graph_noise")
    k = int(g["retrieve_num"])
    e, l, idx = pipeline.retrieve(g["Q"], g["keys"], g["values"], g["labels"], 2 * k)
    assert np.array_equal(l, g["rag_labels"])
    torch.manual_seed(int(g["seed"]))
    noise = torch.normal(mean=0, std=float(g["noise_std"]), size=e.shape).numpy()
    assert np.array_equal(e + noise, g["rag_embeddings"])


def test_g12_edge_large_k():
    """Vanilla-phase retrieval with retrieve_num = 1000 (modules/RAGraph.py:57,73,308-321): the oracle's top-k SET +
    mean against the reference's generate() on the rows whose k-th / (k+1)-th scores differ."""
    g = gold("g12_edge_large_k")
    all_emb = np.concatenate([g["user_embedding"], g["item_embedding"]])
    out, idx, layers, tn, csr = pipeline.edge_forward(g["edges"], g["edge_norm"], g["edge_times"], all_emb, g["resource_keys"],
                                                      g["resource_values"], int(g["retrieve_num"]), float(g["retrieve_weight"]),
                                                      int(g["num_layers"]))
    ref = np.concatenate([g["user_out"], g["item_out"]])
    ok = g["boundary_gap"] > 1e-6
    assert ok.mean() > 0.95
    assert np.allclose(out[ok], ref[ok], atol=2e-5)
    assert idx.shape == (all_emb.shape[0], int(g["retrieve_num"])) and (np.diff(idx, axis=1) > 0).all()
    # the set is the canonical one: k-th score = the k-th largest, every member scores >= it
    S = cref.linear(cref.normalize_rows(all_emb), cref.normalize_rows(g["resource_keys"]))
    kth, idx2 = cref.topk_select_rows(S[:7], 1000)
    for b in range(7):
        assert kth[b] == np.sort(S[b])[::-1][999] and S[b, idx2[b]].min() == kth[b]


def test_g13_bank_build_deterministic_parts():
    """InverseSampling (dense + sparse) and the position codes of a sampled toy graph: oracle vs the reference."""
    g = gold("g13_bank_build")
    for tag in ("adj_norm", "adj_rewired"):
        prob, p, it = cref.compute_sample_prob_dense(g[tag])
        assert np.allclose(p, g[tag + "_pagerank"], atol=2e-6) and 1 <= int(it[0]) < 128
        assert np.allclose(prob, g[tag + "_sample_prob"], rtol=2e-5, atol=1e-7)
    rt, ct, vt = cref.dense_to_csr_t(g["adj_norm"])
    assert np.allclose(cref.csr_row_sums(rt, vt) / np.float32(g["adj_norm"].shape[0] - 1), g["adj_norm_degree_centrality"], atol=1e-7)
    # sparse (edge) flavour: the bi-normalised bipartite adjacency as a COO list
    n = 80
    r, c = g["edge_adj_indices"]
    dense = np.zeros((n, n), np.float32)
    dense[r, c] = g["edge_adj_values"]
    prob, p, it = cref.compute_sample_prob_dense(dense)
    assert np.allclose(p, g["edge_pagerank"], atol=2e-6)
    assert np.allclose(prob, g["edge_sample_prob"], rtol=2e-5, atol=1e-7)
    # position codes
    codes = cref.position_codes_batch(g["sample_adj"][None], g["anchors"][None])[0]
    assert np.allclose(codes, g["position_codes"], atol=1e-6)
    assert np.array_equal(np.isinf(cref.floyd_warshall(g["sample_adj"])), np.isinf(g["sample_dist"]))


def test_g14_ingestion_host_restatement(tmp_path):
    """The host-side restatements of the ingestion (the checkers of the HIP ingestion kernels) against the reference's own
    process_tu_dataset / EdgeListData + _make_binorm_adj outputs."""
    from ragraph_amd.edge_data import EdgeListData
    from ragraph_amd.graph import CSRGraph

    g = gold("g14_ingestion")
    n = g["tu_x"].shape[0]
    csr = CSRGraph.from_edge_index_sym_normalized(torch.from_numpy(g["tu_edge_index"]), n)   # CPU tensors: torch restatement
    dense = np.zeros((n, n), np.float32)
    rows = np.repeat(np.arange(n), np.diff(csr.rowptr.numpy()))
    dense[rows, csr.col.numpy()] = csr.val.numpy()
    assert np.allclose(dense, g["tu_adj"], atol=1e-7) and np.array_equal(dense != 0, g["tu_adj"] != 0)
    F = int(g["tu_num_node_attributes"])
    assert np.array_equal(g["tu_x"][:, :F], g["tu_features"]) and np.array_equal(g["tu_x"][:, F:], g["tu_node_labels"])
    tr, te = tmp_path / "train.txt", tmp_path / "test.txt"
    tr.write_text(str(g["edge_train_txt"]))
    te.write_text(str(g["edge_test_txt"]))
    ds = EdgeListData(str(tr), str(te), hour_interval=int(g["edge_hour_interval"]), device="cpu")
    assert ds.num_users == int(g["edge_num_users"]) and ds.num_items == int(g["edge_num_items"])
    assert np.array_equal(ds.edges.numpy(), g["edge_edges"]) and np.array_equal(ds.edge_times.numpy(), g["edge_times"])
    assert np.allclose(ds.edge_norm.numpy(), g["edge_norm"], atol=1e-7)


def test_g15_graph_fewshot_forward():
    g = gold("g15_graph_fewshot")
    p = {k: g[k] for k in ("W0", "b0", "W1", "b1")}
    p["a0"], p["a1"] = float(g["a0"][0]), float(g["a1"][0])
    csr = cref.dense_to_csr(g["adj"])
    out, idx, h = pipeline.graph_fewshot_forward(g["X"], csr, p, g["keys"], g["values"], g["labels"], g["mean_fewshot_logits"],
                                                 int(g["k"]), float(g["retrieve_weight"]), float(g["label_weight"]))
    assert np.allclose(h, g["H"], atol=1e-5)
    assert np.allclose(out, g["logits"], atol=2e-5)
    # the bank rows the reference built from two resource graphs: every node, normalised keys, its graph's label
    for tag, lo, hi, lab in (("res0", 0, 17, 0), ("res1", 17, 40, 1)):
        hk = pipeline.gcn_layer(g[tag + "_x"], cref.dense_to_csr(g[tag + "_adj"]), p["W0"], p["b0"], p["a0"])
        assert np.allclose(cref.normalize_rows(hk), g["built_keys"][lo:hi], atol=1e-5)
        assert (g["built_labels"][lo:hi].argmax(1) == lab).all()
