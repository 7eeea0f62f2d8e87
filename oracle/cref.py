"""ctypes/numpy front end of oracle/ragraph_oracle.c (the CPU checker).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
ragraph_amd/ imports this module; the product path has no CPU fallback.

Every function takes/returns numpy arrays (float32 / int64 / int32, C-contiguous) and mirrors one entry point of
include/ragraph_hip.h with the same argument meaning; see ragraph_oracle.c for the reference lines each restates.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libragraph_oracle.so")
_lib = None

ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY, ACT_ELU = 0, 1, 2, 3, 4


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (no GPU needed).  Returns the .so path."""
    src = os.path.join(_HERE, "ragraph_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


_c64 = ctypes.c_int64
_ci = ctypes.c_int
_cf = ctypes.c_float


def normalize_rows(X):
    X = _f32(X)
    n, D = X.shape
    out = np.empty_like(X)
    lib().oracle_normalize_rows(_p(X), _c64(n), _ci(D), _p(out))
    return out


def cosine_scores(Qn, Kn):
    Qn, Kn = _f32(Qn), _f32(Kn)
    B, D = Qn.shape
    N = Kn.shape[0]
    S = np.empty((B, N), dtype=np.float32)
    lib().oracle_cosine_scores(_p(Qn), _c64(B), _p(Kn), _c64(N), _ci(D), _p(S))
    return S


def topk_cosine(Q, Kn, k, idx_base=0):
    """Q raw [B,D]; Kn row-normalised [N,D] -> (scores [B,k] f32, idx [B,k] i64), canonical order."""
    Q, Kn = _f32(Q), _f32(Kn)
    B, D = Q.shape
    N = Kn.shape[0]
    assert 1 <= k <= N
    s = np.empty((B, k), dtype=np.float32)
    i = np.empty((B, k), dtype=np.int64)
    lib().oracle_topk_cosine(_p(Q), _c64(B), _p(Kn), _c64(N), _ci(D), _ci(k), _c64(idx_base), _p(s), _p(i))
    return s, i


def topk_merge(scores, idx):
    scores, idx = _f32(scores), _i64(idx)
    G, B, k = scores.shape
    s = np.empty((B, k), dtype=np.float32)
    i = np.empty((B, k), dtype=np.int64)
    lib().oracle_topk_merge(_p(scores), _p(idx), _ci(G), _c64(B), _ci(k), _p(s), _p(i))
    return s, i


def dedup_rows(Kn):
    """Groups of BIT-identical rows (ragraph_dedup_rows_f32): (U, largest group, uniq_row [U] = every group's lowest row,
    ascending; group_ptr [U+1] int32; members [N] int32, every group's rows ascending)."""
    Kn = _f32(Kn)
    N = Kn.shape[0]
    rows = np.ascontiguousarray(Kn).view(np.uint32).reshape(N, -1)
    _, first, inverse = np.unique(rows, axis=0, return_index=True, return_inverse=True)
    inverse = inverse.reshape(-1)
    order = np.argsort(first, kind="stable")              # groups numbered by their lowest row
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    gid = rank[inverse]
    U = order.size
    counts = np.bincount(gid, minlength=U)
    group_ptr = np.zeros(U + 1, dtype=np.int32)
    group_ptr[1:] = np.cumsum(counts)
    members = np.argsort(gid, kind="stable").astype(np.int32)   # stable: ascending rows inside a group
    return U, int(counts.max()), first[order].astype(np.int64), group_ptr, members


def topk_expand_groups(scores_u, idx_u, group_ptr, members, k, idx_base=0, idx_base_u=0):
    """ragraph_topk_expand_groups_f32: every listed group's rows with the group's score, sorted in canonical order (score
    descending, row ascending), the first k; padded with -inf / INT64_MAX."""
    scores_u, idx_u = _f32(scores_u), _i64(idx_u)
    B, ku = scores_u.shape
    U = group_ptr.shape[0] - 1
    out_s = np.full((B, k), -np.inf, dtype=np.float32)
    out_i = np.full((B, k), np.iinfo(np.int64).max, dtype=np.int64)
    for b in range(B):
        cand = []
        for j in range(ku):
            u = int(idx_u[b, j]) - idx_base_u
            if 0 <= u < U:
                cand += [(scores_u[b, j], int(r)) for r in members[group_ptr[u]:group_ptr[u + 1]][:k]]
        cand.sort(key=lambda t: (-t[0], t[1]))
        for r, (sc, row) in enumerate(cand[:k]):
            out_s[b, r], out_i[b, r] = sc, row + idx_base
    return out_s, out_i


def gather_rows(V, idx, idx_base=0):
    V, idx = _f32(V), _i64(idx)
    N, D = V.shape
    out = np.empty(idx.shape + (D,), dtype=np.float32)
    lib().oracle_gather_rows(_p(V), _c64(N), _ci(D), _p(idx), _c64(idx.size), _c64(idx_base), _p(out))
    return out


def gather_reduce(V, L, idx, idx_base=0, v_scale=1.0):
    V, idx = _f32(V), _i64(idx)
    N, D = V.shape
    B, k = idx.shape
    sum_v = np.empty((B, D), dtype=np.float32)
    if L is not None:
        L = _f32(L)
        C = L.shape[1]
        mean_l = np.empty((B, C), dtype=np.float32)
    else:
        C, mean_l = 0, None
    lib().oracle_gather_reduce(_p(V), _ci(D), _p(L), _ci(C), _c64(N), _p(idx), _c64(B), _ci(k), _c64(idx_base),
                               _cf(v_scale), _p(sum_v), _p(mean_l))
    return sum_v, mean_l


def linear(X, W, bias=None, act=ACT_NONE, alpha=0.0):
    X, W = _f32(X), _f32(W)
    M, K = X.shape
    N = W.shape[0]
    assert W.shape[1] == K
    bias = None if bias is None else _f32(bias)
    Y = np.empty((M, N), dtype=np.float32)
    lib().oracle_linear(_p(X), _c64(M), _ci(K), _p(W), _c64(N), _p(bias), _ci(act), _cf(alpha), _p(Y))
    return Y


def spmm_csr(rowptr, col, val, X, bias=None, act=ACT_NONE, alpha=0.0, beta=0.0, Y_in=None):
    rowptr, col, val, X = _i64(rowptr), _i32(col), _f32(val), _f32(X)
    n = rowptr.shape[0] - 1
    D = X.shape[1]
    bias = None if bias is None else _f32(bias)
    Y_in = None if Y_in is None else _f32(Y_in)
    Y = np.empty((n, D), dtype=np.float32)
    lib().oracle_spmm_csr(_p(rowptr), _p(col), _p(val), _c64(n), _p(X), _ci(D), _p(bias), _ci(act), _cf(alpha),
                          _cf(beta), _p(Y_in), _p(Y))
    return Y


def csr_row_normalize(rowptr, val):
    rowptr, val = _i64(rowptr), _f32(val)
    out = np.empty_like(val)
    lib().oracle_csr_row_normalize(_p(rowptr), _p(val), _c64(rowptr.shape[0] - 1), _p(out))
    return out


def segment_softmax(rowptr, x):
    rowptr, x = _i64(rowptr), _f32(x)
    out = np.zeros_like(x)
    lib().oracle_segment_softmax(_p(rowptr), _p(x), _c64(rowptr.shape[0] - 1), _p(out))
    return out


def segment_reduce(X, seg_ptr, w=None, mean_mode=False):
    X, seg_ptr = _f32(X), _i64(seg_ptr)
    D = X.shape[1]
    G = seg_ptr.shape[0] - 1
    w = None if w is None else _f32(w)
    out = np.empty((G, D), dtype=np.float32)
    lib().oracle_segment_reduce(_p(X), _ci(D), _p(seg_ptr), _c64(G), _p(w), _ci(int(mean_mode)), _p(out))
    return out


def position_codes_csr(rowptr, col, val, anchors, dis_q=10.0):
    """(codes [n,A], dist [n,A]) from distances to the anchors only (oracle_position_codes_csr)."""
    rowptr, col, val, anchors = _i64(rowptr), _i32(col), _f32(val), _i64(anchors).reshape(-1)
    n, A = rowptr.shape[0] - 1, anchors.shape[0]
    codes = np.empty((n, A), dtype=np.float32)
    dist = np.empty((n, A), dtype=np.float32)
    lib().oracle_position_codes_csr(_p(rowptr), _p(col), _p(val), _c64(n), _p(anchors), _ci(A), _cf(dis_q), _p(codes), _p(dist))
    return codes, dist


def mul_cols(x, w, act=ACT_NONE, alpha=0.0):
    x, w = _f32(x), _f32(w).reshape(-1)
    out = np.empty_like(x)
    lib().oracle_mul_cols_act(_p(x), _p(w), _c64(x.shape[0]), _ci(x.shape[1]), _ci(act), _cf(alpha), _p(out))
    return out


def axpby(a, wa, b, wb):
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().oracle_axpby(_p(a), _cf(wa), _p(b), _cf(wb), _c64(a.size), _p(out))
    return out


def softmax_mix(logits, rag_label, lam, log_mode=False):
    logits = _f32(logits)
    B, C = logits.shape
    rag_label = None if rag_label is None else _f32(rag_label)
    out = np.empty_like(logits)
    lib().oracle_softmax_mix(_p(logits), _p(rag_label), _c64(B), _ci(C), _cf(lam), _ci(int(log_mode)), _p(out))
    return out


def proto_cosine(emb, proto, mode=0):
    emb, proto = _f32(emb), _f32(proto)
    G, D = emb.shape
    C = proto.shape[0]
    out = np.empty((G, C), dtype=np.float32)
    lib().oracle_proto_cosine(_p(emb), _c64(G), _ci(D), _p(proto), _ci(C), _ci(mode), _p(out))
    return out


# ---- graph helpers restated from the reference's host preprocessing (numpy; small cases) ---------------------------
def dense_to_csr(adj):
    """Row-major non-zeros of a dense adjacency -> (rowptr i64, col i32, val f32); CSR order = ascending column,
    which is the order a dense row-times-matrix product visits the non-zeros."""
    adj = np.asarray(adj, dtype=np.float32)
    n = adj.shape[0]
    rows, cols = np.nonzero(adj)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, cols.astype(np.int32), adj[rows, cols].astype(np.float32)


def coo_to_csr_by_dst(src, dst, n):
    """Edge list -> CSR over destinations, STABLE in the original edge order (the order scatter_add_ accumulates in,
    RAGraph_edge/modules/utils.py:17-32).  Returns (rowptr, col=src, perm) with perm = edge ids in CSR order."""
    src, dst = np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64)
    perm = np.argsort(dst, kind="stable")
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, dst + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, src[perm].astype(np.int32), perm


def topk_rows(S, k):
    S = _f32(S)
    B, N = S.shape
    s = np.empty((B, k), dtype=np.float32)
    i = np.empty((B, k), dtype=np.int64)
    lib().oracle_topk_rows(_p(S), _c64(B), _c64(N), _c64(N), _ci(k), _p(s), _p(i))
    return s, i


def floyd_warshall(adj):
    adj = _f32(adj)
    n = adj.shape[0]
    d = np.empty_like(adj)
    lib().oracle_floyd_warshall(_p(adj), _ci(n), _p(d))
    return d


def position_code(dist, anchors, dis_q=10.0):
    dist, anchors = _f32(dist), _i64(anchors)
    n, A = dist.shape[0], anchors.shape[0]
    out = np.empty((n, A), dtype=np.float32)
    lib().oracle_position_code(_p(dist), _ci(n), _p(anchors), _ci(A), _cf(dis_q), _p(out))
    return out


def topk_select_rows(S, k):
    """Large k (RAGraph_edge/modules/RAGraph.py:57,73,308-321): the canonical top-k SET of every row -- score descending,
    index ascending; of the scores equal to the k-th the lowest indices -- returned in ASCENDING index order, with the
    k-th largest score.  numpy restatement (byte/index work): a stable sort by descending score IS the canonical order."""
    S = _f32(S)
    B, N = S.shape
    kth = np.empty(B, dtype=np.float32)
    idx = np.empty((B, k), dtype=np.int64)
    for b in range(B):
        order = np.argsort(-S[b], kind="stable")[:k]
        kth[b] = S[b, order[-1]]
        idx[b] = np.sort(order)
    return kth, idx


# ---- toy-bank construction --------------------------------------------------------------
def csr_row_sums(rowptr, val):
    rowptr, val = _i64(rowptr), _f32(val)
    out = np.empty(rowptr.shape[0] - 1, dtype=np.float32)
    lib().oracle_csr_row_sums(_p(rowptr), _p(val), _c64(out.shape[0]), _p(out))
    return out


def pagerank(rowptr_t, col_t, val_t, out_deg, graph_ptr, d=0.85, eps=1e-6, max_iter=128):
    """InverseSampling.pagerank_algorithm over a batch of graphs (transposed CSR, see ragraph_hip.h) -> (p, iters)."""
    rowptr_t, col_t, val_t, out_deg, graph_ptr = _i64(rowptr_t), _i32(col_t), _f32(val_t), _f32(out_deg), _i64(graph_ptr)
    p = np.empty(out_deg.shape[0], dtype=np.float32)
    iters = np.empty(graph_ptr.shape[0] - 1, dtype=np.int32)
    lib().oracle_pagerank(_p(rowptr_t), _p(col_t), _p(val_t), _p(out_deg), _p(graph_ptr), _c64(iters.shape[0]), _cf(d), _cf(eps),
                          _ci(max_iter), _p(p), _p(iters))
    return p, iters


def sample_prob(pagerank_p, col_sum, graph_ptr, alpha=0.5, eps=1e-6):
    pr, cs, graph_ptr = _f32(pagerank_p), _f32(col_sum), _i64(graph_ptr)
    out = np.empty_like(pr)
    lib().oracle_sample_prob(_p(pr), _p(cs), _p(graph_ptr), _c64(graph_ptr.shape[0] - 1), _cf(alpha), _cf(eps), _p(out))
    return out


def dense_to_csr_t(adj):
    """CSR of the TRANSPOSED dense matrix (row j lists the i with adj[i][j] != 0, ascending i) + the row sums of adj."""
    a = _f32(adj)
    rowptr, col, val = dense_to_csr(np.ascontiguousarray(a.T))
    return rowptr, col, val


def compute_sample_prob_dense(adj):
    """InverseSampling.compute_sample_prob(adj) for one dense adjacency: (prob, pagerank, iters)."""
    a = _f32(adj)
    n = a.shape[0]
    rowptr, col, val = dense_to_csr(a)
    rt, ct, vt = dense_to_csr_t(a)
    gp = np.array([0, n], dtype=np.int64)
    out_deg = csr_row_sums(rowptr, val)
    p, it = pagerank(rt, ct, vt, out_deg, gp)
    return sample_prob(p, csr_row_sums(rt, vt), gp), p, it


def position_codes_batch(adj, anchors, dis_q=10.0):
    """PositionAwareEncoder.encode_position_aware_code for G small graphs: adj [G,n,n], anchors [G,A] -> [G,n,A]."""
    adj, anchors = _f32(adj), _i64(anchors)
    return np.stack([position_code(floyd_warshall(adj[g]), anchors[g], dis_q) for g in range(adj.shape[0])])
