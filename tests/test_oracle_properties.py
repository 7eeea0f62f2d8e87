"""Property tests of the CPU oracle itself (hypothesis, small shapes): the checker must be right before it checks.
Independent numpy / scipy formulations, including ragged and degenerate inputs (duplicate keys, zero rows, empty CSR
rows, k = N, single element)."""
import numpy as np
import scipy.sparse as sp
from hypothesis import given, settings, strategies as st

from oracle import cref

SET = settings(max_examples=40, deadline=None)


def _canonical_topk(S, k):
    idx = np.empty((S.shape[0], k), dtype=np.int64)
    for b in range(S.shape[0]):
        order = np.lexsort((np.arange(S.shape[1]), -S[b].astype(np.float64)))  # score desc, index asc
        idx[b] = order[:k]
    return idx


@SET
@given(B=st.integers(1, 9), N=st.integers(1, 70), D=st.sampled_from([1, 3, 8, 64]), seed=st.integers(0, 10_000),
       dup=st.booleans(), data=st.data())
def test_topk_cosine_is_canonical_selection_of_its_own_scores(B, N, D, seed, dup, data):
    k = data.draw(st.integers(1, N))
    rng = np.random.default_rng(seed)
    keys = rng.standard_normal((N, D)).astype(np.float32)
    if dup and N > 2:
        keys[N // 2:] = keys[: N - N // 2]          # exact duplicate keys, as toy banks have
    kn = cref.normalize_rows(keys)
    q = rng.standard_normal((B, D)).astype(np.float32)
    if B > 1:
        q[0] = 0                                     # zero-norm query: every score is 0
    s, i = cref.topk_cosine(q, kn, k, idx_base=3)
    S = cref.cosine_scores(cref.normalize_rows(q), kn)
    ref_i = _canonical_topk(S, k)
    assert np.array_equal(i - 3, ref_i)
    assert np.array_equal(s, np.take_along_axis(S, ref_i, 1))
    # sharding the bank and merging gives the same answer
    if N >= 2 and k <= N // 2:
        h = N // 2
        parts = [cref.topk_cosine(q, kn[:h], k, 0), cref.topk_cosine(q, kn[h:], k, h)]
        ms, mi = cref.topk_merge(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]))
        assert np.array_equal(mi, ref_i) and np.array_equal(ms, s)
    assert np.array_equal(cref.topk_rows(S, k)[1], ref_i)


@SET
@given(n=st.integers(1, 40), D=st.sampled_from([4, 8, 20, 256]), seed=st.integers(0, 10_000), act=st.integers(0, 3))
def test_spmm_matches_scipy(n, D, seed, act):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=0.15, random_state=seed, format="csr", dtype=np.float32)
    A.sort_indices()
    X = rng.standard_normal((n, D)).astype(np.float32)
    b = rng.standard_normal(D).astype(np.float32)
    Y = cref.spmm_csr(A.indptr, A.indices, A.data, X, bias=b, act=act, alpha=0.25)
    ref = (A.astype(np.float64) @ X.astype(np.float64)) + b
    ref = {0: ref, 1: np.maximum(ref, 0), 2: np.where(ref >= 0, ref, 0.25 * ref), 3: np.where(ref >= 0, ref, 0.25 * ref)}[act]
    assert np.allclose(Y, ref, atol=1e-4)
    rn = cref.csr_row_normalize(A.indptr.astype(np.int64), A.data)
    sums = np.add.reduceat(rn, A.indptr[:-1][np.diff(A.indptr) > 0]) if A.nnz else np.zeros(0)
    assert np.allclose(sums, 1.0, atol=1e-5)


@SET
@given(n=st.integers(1, 30), D=st.integers(1, 300), seed=st.integers(0, 10_000))
def test_normalize_rows_is_f_normalize(n, D, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, D)).astype(np.float32) * rng.choice([1e-20, 1e-3, 1.0, 1e6])
    x[0] = 0
    out = cref.normalize_rows(x)
    nrm = np.maximum(np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True), 1e-12)
    assert np.allclose(out, x / nrm, rtol=2e-6, atol=1e-30)
    assert not np.isnan(out).any()


@SET
@given(M=st.integers(1, 20), K=st.integers(1, 70), N=st.integers(1, 20), seed=st.integers(0, 10_000))
def test_linear_matches_numpy(M, K, N, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    assert np.allclose(cref.linear(X, W, b), X.astype(np.float64) @ W.astype(np.float64).T + b, atol=1e-4)
