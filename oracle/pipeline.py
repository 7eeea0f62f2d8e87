"""Whole-forward CPU restatements built from oracle/cref.py (the C checker).  TEST INFRASTRUCTURE ONLY.

Each function follows one reference forward line by line (citations relative to the reference root) with the two
structural changes the HIP path also makes, neither of which changes a result:
  * the adjacency is CSR (non-zeros in ascending column order = the order a dense product visits them) instead of the
    reference's dense n x n matrix (layers/gcn.py:36, Propagation.py:15-22);
  * keys are normalised once per bank instead of on every call (SimilarityFunctions.py:11).
PrePrompt.embed's get_subgraph_3 loop (preprompt.py:8-27,60) is not restated: inference() discards its result.
"""
from __future__ import annotations

import numpy as np

from . import cref


def aggregate_first_applies(f_in, f_out):
    """The shapes for which the HIP path's INFERENCE evaluates a layer as (A_hat X) W^T (ragraph_amd/layers/gcn.py)."""
    return f_in % 4 == 0 and f_in >= 16 and 2 * f_in <= f_out


def gcn_layer(X, csr, W, bias, alpha, order="reference"):
    """layers/gcn.py:26-40: PReLU_alpha(A_hat @ (X W^T) + b) -- the REFERENCE's association, always, unless the caller
    asks for the other one: order="aggregate_first" gives (A_hat @ X) W^T + b for the shapes the HIP path's inference
    re-associates (aggregate_first_applies; every other shape: the reference order).  The two differ by fp32 rounding
    (<= 1e-5 on the fixtures); the HIP path is held bit for bit to the reference order under
    RAGRAPH_GCN_REFERENCE_ORDER=1 and to the re-associated form otherwise, and its default output to the reference order
    within 1e-5 (tests/test_gpu_models.py, tests/test_gpu_fullsize.py)."""
    rowptr, col, val = csr
    X, W = np.asarray(X, dtype=np.float32), np.asarray(W, dtype=np.float32)
    if order not in ("reference", "aggregate_first"):
        raise ValueError(order)
    if order == "aggregate_first" and aggregate_first_applies(X.shape[1], W.shape[0]):
        return cref.linear(cref.spmm_csr(rowptr, col, val, X), W, bias, act=cref.ACT_PRELU, alpha=float(alpha))
    return cref.spmm_csr(rowptr, col, val, cref.linear(X, W), bias=bias, act=cref.ACT_PRELU, alpha=float(alpha))


def propagate(csr, x, k):
    """Propagation.py:7-27: A_tilde = A / A.sum(1); k x { x = relu(A_tilde @ x) }."""
    rowptr, col, val = csr
    valn = cref.csr_row_normalize(rowptr, val)
    for _ in range(int(k)):
        x = cref.spmm_csr(rowptr, col, valn, x, act=cref.ACT_RELU)
    return x


def retrieve(search_keys, keys, values, labels, k):
    """ToyGraphBase.py:47-81 without noise: (rag_embeddings [B,k,D], rag_labels [B,k,C], idx [B,k])."""
    q = np.asarray(search_keys, dtype=np.float32)
    q2 = q.reshape(1, -1) if q.ndim == 1 else q  # graph flavour: 1-D query, scores unsqueezed (graph ToyGraphBase.py:73)
    kn = cref.normalize_rows(keys)
    _, idx = cref.topk_cosine(q2, kn, int(k))
    return cref.gather_rows(values, idx), (None if labels is None else cref.gather_rows(labels, idx)), idx


def decoder(x, fc1_w, fc1_b, fc2_w, fc2_b):
    """TaskDecoder.py:14-17: fc2(LeakyReLU_0.01(fc1(x)))."""
    return cref.linear(cref.linear(x, fc1_w, fc1_b, act=cref.ACT_LEAKY, alpha=0.01), fc2_w, fc2_b)


def node_forward(X, csr, p, keys, values, labels, k, hops, retrieve_weight, label_weight):
    """RAGraph_node/RAGraph.py:39-59 (finetune branch).  p: dict W,bias,alpha,fc1_w,fc1_b,fc2_w,fc2_b."""
    h = gcn_layer(X, csr, p["W"], p["bias"], p["alpha"])                      # :40 pretrain_model.inference
    kn = cref.normalize_rows(keys)
    _, idx = cref.topk_cosine(h, kn, int(k))                                   # :43 retrieve
    rag_emb, rag_label = cref.gather_reduce(values, labels, idx)               # :48-49 mean labels, sum values
    q = propagate(csr, h, hops)                                                # :51
    hidden = cref.axpby(q, np.float32(1.0 - retrieve_weight), rag_emb, np.float32(retrieve_weight))  # :53
    logits = decoder(hidden, p["fc1_w"], p["fc1_b"], p["fc2_w"], p["fc2_b"])   # :54
    return cref.softmax_mix(logits, rag_label, float(label_weight)), idx, h    # :55-57


def graph_forward(X, csr, p, keys, values, labels, k, hops, retrieve_weight, label_weight):
    """RAGraph_graph/RAGraph.py:48-71: one pooled query per graph."""
    n = X.shape[0]
    seg = np.array([0, n], dtype=np.int64)
    h = gcn_layer(X, csr, p["W"], p["bias"], p["alpha"])                      # :49
    g = cref.segment_reduce(h, seg, mean_mode=True)                            # :50 mean over nodes -> [1,D]
    kn = cref.normalize_rows(keys)
    _, idx = cref.topk_cosine(g, kn, int(k))                                   # :53
    rag_emb, rag_label = cref.gather_reduce(values, labels, idx)               # :59-60
    q = cref.segment_reduce(propagate(csr, h, hops), seg, mean_mode=True)      # :62-63
    hidden = cref.axpby(q, np.float32(1.0 - retrieve_weight), rag_emb, np.float32(retrieve_weight))  # :65
    logits = decoder(hidden, p["fc1_w"], p["fc1_b"], p["fc2_w"], p["fc2_b"])
    return cref.softmax_mix(logits, rag_label, float(label_weight)), idx, h


def edge_time_norm(rowptr, perm, edge_times, max_step=None):
    """RAGraph_edge/modules/RAGraph.py:250-263: min-max scale, scatter_softmax per destination.  Returns values in CSR
    (destination-sorted) order."""
    t = np.asarray(edge_times).astype(np.float32)
    tmin = t.min()
    mx = np.float32(t.max() if max_step is None else max_step)
    t = (t - tmin) / (mx - tmin)
    return cref.segment_softmax(rowptr, t[perm])


def edge_forward(edges, edge_norm, edge_times, gated_emb, resource_keys, resource_values, k, retrieve_weight,
                 num_layers=3, max_step=None):
    """RAGraph_edge/modules/RAGraph.py:265-333 from the gated layer-0 embeddings on (LoRA / gating are training-side)."""
    n = gated_emb.shape[0]
    rowptr, col, perm = cref.coo_to_csr_by_dst(edges[:, 0], edges[:, 1], n)   # _agg scatters by edges[:,1] (:236-239)
    tn = edge_time_norm(rowptr, perm, edge_times, max_step)                    # :266
    # :267  edge_norm * 1/2 + time_norm * 1/2  (halving is exact in fp32, so this is one rounded add)
    norm = cref.axpby(np.asarray(edge_norm, dtype=np.float32)[perm], 0.5, tn, 0.5)
    layers = [np.asarray(gated_emb, dtype=np.float32)]
    for _ in range(int(num_layers)):                                           # :280-283
        layers.append(cref.spmm_csr(rowptr, col, norm, layers[-1]))
    kn = cref.normalize_rows(resource_keys)
    if int(k) <= 64:
        _, idx = cref.topk_cosine(layers[0], kn, int(k))                       # :298-311 (slabs do not change results)
    else:  # vanilla phase: retrieve_num in the thousands (:57,73) -- only the winners' mean is consumed: the SET suffices
        _, idx = cref.topk_select_rows(cref.linear(cref.normalize_rows(layers[0]), kn), int(k))
    rag, _ = cref.gather_reduce(resource_values, None, idx, v_scale=np.float32(1.0 / k))  # :314,321 mean over k
    acc = layers[0]
    for l in layers[1:]:                                                       # :327 sum(res_emb)
        acc = cref.axpby(acc, 1.0, l, 1.0)
    out = cref.axpby(acc, np.float32(1.0 - retrieve_weight), rag, np.float32(retrieve_weight))  # :328
    return out, idx, layers, tn, (rowptr, col, perm)


def downprompt_logits(h, w, graph_len, proto, log_softmax=True):
    """RAGraph_graph/downprompt.py:154-168 (w*h), :98-112 (per-graph sum), :41-56 (cosine to class means, log_softmax)."""
    seg = np.concatenate([[0], np.cumsum(np.asarray(graph_len))]).astype(np.int64)
    emb = cref.segment_reduce(h, seg, w=np.asarray(w, dtype=np.float32).reshape(-1))
    return cref.proto_cosine(emb, proto, mode=2 if log_softmax else 0), emb


def downprompt_node_averageemb(labels, rawret):
    """RAGraph_node/downprompt.py:59-78 with a zero-filled buffer: class sums (rows in index order) / floor(n/2)."""
    labels = np.asarray(labels).reshape(-1)
    rawret = np.asarray(rawret, dtype=np.float32)
    order = np.concatenate([np.nonzero(labels == c)[0] for c in range(3)])
    seg = np.concatenate([[0], np.cumsum([(labels == c).sum() for c in range(3)])]).astype(np.int64)
    sums = cref.segment_reduce(rawret[order], seg)
    half = int(rawret.shape[0] / 2)
    return cref.mul_cols(sums, np.full(rawret.shape[1], np.float32(1.0) / np.float32(half), dtype=np.float32))


def downprompt_node_forward(h, w, ave):
    """RAGraph_node/downprompt.py:25-46: ELU(w * h) (:118-130) -> cosine to ave[0..2] -> softmax(dim=1)."""
    rawret = cref.mul_cols(h, w, cref.ACT_ELU, 1.0)
    return cref.proto_cosine(rawret, ave, mode=1), rawret


def fewshot_scores(search_keys, adj_dense, anchors, keys, positions, structure_weight=0.001, semantic_weight=0.999,
                   dis_q=10.0, all_pairs=False):
    """RAGraph_node_fewshot/ragraph_utils/ToyGraphBase.py:47-61: w_s * cos(position codes) + w_m * cos(embeddings).
    Codes (PositionAwareEncoder.py:6-24): all_pairs=True through the reference's Floyd-Warshall matrix; default through
    the distances to the anchors only (oracle_position_codes_csr: what the product computes per forward; the two agree
    to ~1 ulp of a path sum -- tests/test_oracle_golden.py pins both against the reference's codes)."""
    if all_pairs:
        pos = cref.position_code(cref.floyd_warshall(adj_dense), anchors, dis_q)
    else:
        pos, _ = cref.position_codes_csr(*cref.dense_to_csr(adj_dense), anchors, dis_q)
    s_struct = cref.linear(cref.normalize_rows(pos), cref.normalize_rows(positions))
    s_sem = cref.linear(cref.normalize_rows(search_keys), cref.normalize_rows(keys))
    return cref.axpby(s_struct, structure_weight, s_sem, semantic_weight), pos


def fewshot_retrieve(search_keys, adj_dense, anchors, keys, values, labels, positions, k):
    """ToyGraphBase.py:47-79 without noise."""
    scores, pos = fewshot_scores(search_keys, adj_dense, anchors, keys, positions)
    _, idx = cref.topk_rows(scores, int(k))
    return cref.gather_rows(values, idx), cref.gather_rows(labels, idx), idx, pos


def edge_topk_items(user_emb, item_emb, users, hist, k=20):
    """RAGraph_edge/utils/metrics.py:104-116: rating, history mask (-1e8), top-k.  hist: list of item-id lists."""
    rating = cref.linear(np.asarray(user_emb, dtype=np.float32)[users], item_emb)
    for r, items in enumerate(hist):
        rating[r, list(items)] = np.float32(-1e8)
    return cref.topk_rows(rating, k)[1]


def graph_fewshot_forward(X, csr, p, keys, values, labels, mean_fewshot_logits, k, retrieve_weight, label_weight, hops=1):
    """RAGraph_graph_fewshot/RAGraph.py:46-86 (finetune branch, no noise).  p: W0,b0,a0 (encode = layer 0), W1,b1,a1
    (decode = layer 1)."""
    h = gcn_layer(X, csr, p["W0"], p["b0"], p["a0"])                              # :47 encode
    kn = cref.normalize_rows(keys)
    _, idx = cref.topk_cosine(h, kn, int(k))                                       # :51
    label_ids = np.argmax(cref.gather_rows(labels, idx), axis=-1).astype(np.int64)  # :55
    rag_logits, _ = cref.gather_reduce(mean_fewshot_logits, None, label_ids, v_scale=np.float32(1.0 / k))   # :56,68
    rag_emb, _ = cref.gather_reduce(values, None, idx)                             # :69
    q = propagate(csr, h, hops)                                                    # :71
    hidden = cref.axpby(q, np.float32(1.0 - retrieve_weight), rag_emb, np.float32(retrieve_weight))          # :75
    dec = gcn_layer(hidden, csr, p["W1"], p["b1"], p["a1"])                        # :79 decode
    mix = cref.axpby(dec, np.float32(1.0 - label_weight), rag_logits, np.float32(label_weight))              # :82
    seg = np.array([0, mix.shape[0]], dtype=np.int64)
    return cref.segment_reduce(mix, seg, mean_mode=True), idx, h                   # :84
