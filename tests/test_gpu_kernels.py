"""HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar: BIT-EXACT for every fmaf-chain kernel (scores, linear, SpMM, norms, gathers, sums) and for all indices; 1e-6
for the kernels that call expf/logf (softmax family), whose libm differs between host and device.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rng(seed):
    return np.random.default_rng(seed)


def _bank(rng, N, D):
    return cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))


@pytest.fixture(params=["fused", "slab-rule"])
def topk_family(request, monkeypatch):
    """Small score matrices take the slab path (dense kernel + topk_rows) by default; RAGRAPH_TOPK_SLAB=0 (read per
    call) keeps every shape on the fused kernels.  The fp32 top-k tests run under both, so that neither family of
    kernels loses its small-shape coverage to the dispatch rule."""
    if request.param == "fused":
        monkeypatch.setenv("RAGRAPH_TOPK_SLAB", "0")
    else:
        monkeypatch.delenv("RAGRAPH_TOPK_SLAB", raising=False)
    return request.param


@pytest.mark.parametrize("n,D", [(1, 1), (5, 3), (7, 64), (33, 128), (257, 256), (4, 1433), (3, 300)])
def test_normalize_rows_bit_exact(dev, n, D):
    from ragraph_amd import kernels as K

    x = _rng(n * 1000 + D).standard_normal((n, D), dtype=np.float32)
    x[0] *= 1e-3
    if n > 1:
        x[1] = 0.0  # zero row: eps clamp -> zeros (F.normalize eps semantics)
    got = K.normalize_rows(_t(x, dev)).cpu().numpy()
    ref = cref.normalize_rows(x)
    assert np.array_equal(got, ref)
    # and it is F.normalize up to rounding of the norm
    tref = torch.nn.functional.normalize(torch.from_numpy(x), p=2, dim=-1).numpy()
    assert np.allclose(got, tref, atol=1e-6)


@pytest.mark.parametrize(
    "B,N,D,k",
    [
        (1, 1000, 256, 3),       # graph flavour: one query
        (7, 33, 64, 5),          # tiny bank, ragged tile
        (64, 4096, 64, 10),
        (300, 5000, 128, 8),     # > one query tile, N not a stage multiple
        (256, 20000, 256, 10),
        (513, 70001, 256, 10),   # several splits, ragged everything
        (40, 10, 256, 10),       # k == N
        (100, 3000, 256, 1),
        (50, 9000, 256, 32),     # k at the fused kernel's maximum
        (300, 9000, 64, 50),     # 32 < k <= 64: materialised slabs + row top-k (edge 'vanilla' retrieve_num = 50)
        (3, 2000, 256, 64),
        (5, 70000, 256, 10),     # small-batch kernel with the sampled-threshold pre-pass (4 <= B <= 16, N >= 64k)
        (16, 131072, 64, 7),
        (3, 70000, 128, 10),     # small-batch kernel without the pre-pass
        (100, 70000, 256, 10),   # streaming kernel, 7 groups of 16 queries (last one ragged), with pre-pass
        (128, 100000, 64, 5),    # 8 groups (largest batch the streaming kernel takes)
        (40, 3000, 256, 31),     # largest k the streaming kernel's LDS holds
    ],
)
def test_topk_cosine_bit_exact(dev, topk_family, B, N, D, k):
    from ragraph_amd import kernels as K

    rng = _rng(B * 7 + N + D + k)
    kn = _bank(rng, N, D)
    q = rng.standard_normal((B, D), dtype=np.float32)
    s, i = K.topk_cosine(_t(q, dev), _t(kn, dev), k, idx_base=5)
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=5)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(s.cpu().numpy(), rs)


def test_topk_cosine_duplicates_and_zero_query(dev, topk_family):
    """Toy banks hold exact duplicate keys (multinomial with replacement, ToyGraphBase.py:98); torch.topk leaves the
    order of ties open, the library breaks them towards the lower index."""
    from ragraph_amd import kernels as K

    rng = _rng(11)
    base = _bank(rng, 500, 256)
    kn = np.concatenate([base, base[::-1], base[:100]], axis=0)  # every key 2-3 times
    q = rng.standard_normal((70, 256), dtype=np.float32)
    q[3] = 0.0  # zero-norm query: all scores 0 -> lowest k indices
    s, i = K.topk_cosine(_t(q, dev), _t(kn, dev), 10)
    rs, ri = cref.topk_cosine(q, kn, 10)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(s.cpu().numpy(), rs)
    assert np.array_equal(ri[3], np.arange(10))


def test_topk_merge_matches_single_shard(dev):
    from ragraph_amd import kernels as K

    rng = _rng(5)
    N, D, B, k, G = 8000, 128, 77, 10, 4
    kn = _bank(rng, N, D)
    q = rng.standard_normal((B, D), dtype=np.float32)
    qd, knd = _t(q, dev), _t(kn, dev)
    full_s, full_i = K.topk_cosine(qd, knd, k)
    per = N // G
    ss, ii = [], []
    for g in range(G):
        s, i = K.topk_cosine(qd, knd[g * per:(g + 1) * per].contiguous(), k, idx_base=g * per)
        ss.append(s)
        ii.append(i)
    ms, mi = K.topk_merge(torch.stack(ss), torch.stack(ii))
    assert torch.equal(mi, full_i) and torch.equal(ms, full_s)
    rs, ri = cref.topk_merge(torch.stack(ss).cpu().numpy(), torch.stack(ii).cpu().numpy())
    assert np.array_equal(mi.cpu().numpy(), ri) and np.array_equal(ms.cpu().numpy(), rs)


def test_gather_rows_and_reduce(dev):
    from ragraph_amd import kernels as K

    rng = _rng(9)
    N, D, C, B, k = 1000, 256, 3, 37, 10
    V = rng.standard_normal((N, D), dtype=np.float32)
    L = np.eye(C, dtype=np.float32)[rng.integers(0, C, N)]
    idx = rng.integers(0, N, (B, k))
    assert np.array_equal(K.gather_rows(_t(V, dev), _t(idx, dev)).cpu().numpy(), V[idx])
    sv, ml = K.gather_reduce(_t(V, dev), _t(L, dev), _t(idx, dev))
    rsv, rml = cref.gather_reduce(V, L, idx)
    assert np.array_equal(sv.cpu().numpy(), rsv) and np.array_equal(ml.cpu().numpy(), rml)
    assert np.allclose(rsv, V[idx].sum(1), atol=1e-5) and np.allclose(rml, L[idx].mean(1), atol=1e-6)
    # shard semantics: winners outside [base, base+N) contribute zero; two half-shards sum to the whole
    h = N // 2
    a, _ = K.gather_reduce(_t(V[:h], dev), None, _t(idx, dev), idx_base=0)
    b, _ = K.gather_reduce(_t(V[h:], dev), None, _t(idx, dev), idx_base=h)
    ra, _ = cref.gather_reduce(V[:h], None, idx, idx_base=0)
    assert np.array_equal(a.cpu().numpy(), ra)
    assert np.allclose((a + b).cpu().numpy(), rsv, atol=1e-5)
    # edge flavour: mean of values (v_scale = 1/k), odd width
    V3 = rng.standard_normal((N, 30), dtype=np.float32)
    m, _ = K.gather_reduce(_t(V3, dev), None, _t(idx, dev), v_scale=1.0 / k)
    rm, _ = cref.gather_reduce(V3, None, idx, v_scale=1.0 / k)
    assert np.array_equal(m.cpu().numpy(), rm)


@pytest.mark.parametrize("D,k,C", [(256, 1, 3), (64, 10, 7), (128, 64, 2), (256, 65, 3), (64, 200, 70), (512, 9, 3)])
def test_gather_reduce_index_blocks(dev, D, k, C):
    """Sum_k V[idx] / mean_k L[idx] with the wave's indices read first and handed out lane by lane (blocks of 64 winners): one
    winner, a full block, one past a block, several blocks, more label columns than lanes, two float4 columns per lane; winners of
    another shard (idx_base) and a ragged last workgroup -- the oracle's bits."""
    from ragraph_amd import kernels as K

    rng = _rng(D + 7 * k + C)
    N, B = 777, 41
    V = rng.standard_normal((N, D), dtype=np.float32)
    L = rng.standard_normal((N, C), dtype=np.float32)
    idx = rng.integers(0, 2 * N, (B, k))            # half of the winners belong to "another shard"
    for base in (0, N):
        sv, ml = K.gather_reduce(_t(V, dev), _t(L, dev), _t(idx, dev), idx_base=base, v_scale=0.5)
        rsv, rml = cref.gather_reduce(V, L, idx, base, 0.5)
        assert np.array_equal(sv.cpu().numpy(), rsv) and np.array_equal(ml.cpu().numpy(), rml)


@pytest.mark.parametrize("M,K_,N_,act", [(4096, 256, 3, 0), (5001, 256, 7, 3), (4100, 32, 1, 2), (9999, 512, 8, 1), (4097, 64, 2, 0)])
def test_linear_narrow_heads_bit_exact(dev, M, K_, N_, act):
    """Tall X against a handful of output columns (TaskDecoder's fc2 on every node of a large graph): the row-streaming kernel
    gives the oracle's fmaf chains bit for bit -- ragged last wave, every width 1..8, bias or none, every epilogue."""
    from ragraph_amd import kernels as K

    rng = _rng(M + K_ + N_)
    X = rng.standard_normal((M, K_), dtype=np.float32)
    W = (rng.standard_normal((N_, K_), dtype=np.float32) * 0.2).astype(np.float32)
    b = rng.standard_normal(N_, dtype=np.float32)
    for bias in (b, None):
        got = K.linear(_t(X, dev), _t(W, dev), None if bias is None else _t(bias, dev), act=act, alpha=0.2).cpu().numpy()
        assert np.array_equal(got, cref.linear(X, W, bias=bias, act=act, alpha=0.2))


@pytest.mark.parametrize("G,m,B,k", [(8, 4, 5001, 10), (8, 2, 4096, 10), (2, 10, 7777, 10), (8, 4, 100, 10), (5, 8, 6000, 10), (4, 16, 5000, 3)])
def test_theta_sharpen_kth_of_the_union(dev, G, m, B, k):
    """theta[b] = max(theta[b], k-th largest of the union of every shard's m scores of query b): one query per wave, and two per
    wave where G m <= 32 and the batch is large -- duplicates count as often as they occur, an odd last query, -inf slots."""
    from ragraph_amd import kernels as K

    rng = _rng(G * 100 + m + B)
    g = rng.standard_normal((G, B, m)).astype(np.float32)
    g[:, ::7, :] = np.round(g[:, ::7, :], 1)            # ties
    g[0, 5, :] = -np.inf
    th = rng.standard_normal(B).astype(np.float32)
    kth = np.sort(g.transpose(1, 0, 2).reshape(B, G * m), axis=1)[:, ::-1][:, k - 1]
    got = K.theta_sharpen(_t(g, dev), _t(th.copy(), dev), k).cpu().numpy()
    assert np.array_equal(got, np.maximum(th, kth))


@pytest.mark.parametrize("D,k", [(256, 10), (64, 70), (30, 5)])
def test_gather_reduce_mix_is_reduce_then_axpby(dev, D, k):
    """ragraph_gather_reduce_mix_f32 = ragraph_gather_reduce_f32 followed by ragraph_axpby_f32, bit for bit (RAGraph.py:48-49 +
    :53 in one launch), label means included; vectorised and scalar widths, winners of another shard."""
    from ragraph_amd import kernels as K

    rng = _rng(D * k)
    N, B, C = 500, 77, 3
    V = rng.standard_normal((N, D), dtype=np.float32)
    L = rng.standard_normal((N, C), dtype=np.float32)
    Aq = rng.standard_normal((B, D), dtype=np.float32)
    idx = rng.integers(0, N + 100, (B, k))
    Vd, Ld, id_, Ad = _t(V, dev), _t(L, dev), _t(idx, dev), _t(Aq, dev)
    for vs in (1.0, 0.25):
        s, ml = K.gather_reduce(Vd, Ld, id_, v_scale=vs)
        two = K.axpby(Ad, 0.3, s, 0.7)
        one, ml1 = K.gather_reduce_mix(Vd, Ld, id_, Ad, 0.3, 0.7, v_scale=vs)
        assert torch.equal(one, two) and torch.equal(ml1, ml)
    one, none = K.gather_reduce_mix(Vd, None, id_, Ad, 0.5, 0.5)
    assert none is None and torch.equal(one, K.axpby(Ad, 0.5, K.gather_reduce(Vd, None, id_)[0], 0.5))


@pytest.mark.parametrize("M,K_,N_,act", [(1, 1, 1, 0), (33, 18, 256, 2), (200, 1433, 256, 0), (130, 256, 3, 3),
                                         (65, 256, 256, 3), (64, 64, 64, 1),
                                         # the 128 x 128 tile kernel (M, N >= 128, K % 4 == 0): ragged tiles, K not a
                                         # multiple of the 32-wide chunk, one chunk, many chunks
                                         (300, 128, 256, 2), (129, 36, 130, 1), (128, 4, 128, 0), (1000, 260, 384, 3),
                                         (257, 1000, 129, 0), (2708, 1433, 128, 2), (31, 257, 17, 3),
                                         # the streaming kernel (M >= 4096, K in {64,128,256}, N in (192,256] per block)
                                         (4096, 128, 256, 2), (4133, 256, 250, 3), (5000, 64, 512, 0), (4100, 128, 700, 1),
                                         (9000, 128, 193, 2)])
def test_linear_bit_exact(dev, M, K_, N_, act):
    from ragraph_amd import kernels as K

    rng = _rng(M + K_ + N_)
    X = rng.standard_normal((M, K_), dtype=np.float32)
    W = (rng.standard_normal((N_, K_), dtype=np.float32) / np.sqrt(K_)).astype(np.float32)
    b = rng.standard_normal(N_, dtype=np.float32)
    got = K.linear(_t(X, dev), _t(W, dev), _t(b, dev), act=act, alpha=0.25).cpu().numpy()
    ref = cref.linear(X, W, b, act=act, alpha=0.25)
    assert np.array_equal(got, ref)
    got = K.linear(_t(X, dev), _t(W, dev)).cpu().numpy()
    assert np.array_equal(got, cref.linear(X, W))
    assert np.allclose(got, X @ W.T, atol=1e-4)


def _rand_csr(rng, n, ncols, mean_deg, empty_rows=True):
    deg = rng.poisson(mean_deg, n)
    if empty_rows and n > 2:
        deg[1] = 0
    if n > 3:
        deg[3] = 5 * mean_deg + 67  # one long row
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, ncols, rowptr[-1]).astype(np.int32)
    val = rng.random(rowptr[-1], dtype=np.float32) + 0.1
    return rowptr, col, val


@pytest.mark.parametrize("n,D,act", [(50, 256, 2), (300, 64, 1), (77, 128, 0), (10, 4, 1), (40, 512, 3), (64, 20, 0)])
def test_spmm_csr_bit_exact(dev, n, D, act):
    from ragraph_amd import kernels as K

    rng = _rng(n + D)
    rowptr, col, val = _rand_csr(rng, n, n, 6)
    X = rng.standard_normal((n, D), dtype=np.float32)
    b = rng.standard_normal(D, dtype=np.float32)
    Yin = rng.standard_normal((n, D), dtype=np.float32)
    args = (_t(rowptr, dev), _t(col, dev), _t(val, dev), _t(X, dev))
    got = K.spmm_csr(*args, bias=_t(b, dev), act=act, alpha=0.25).cpu().numpy()
    assert np.array_equal(got, cref.spmm_csr(rowptr, col, val, X, bias=b, act=act, alpha=0.25))
    got = K.spmm_csr(*args, act=act, alpha=0.25, beta=1.0, y_in=_t(Yin, dev)).cpu().numpy()
    assert np.array_equal(got, cref.spmm_csr(rowptr, col, val, X, act=act, alpha=0.25, beta=1.0, Y_in=Yin))


def test_csr_row_normalize_and_segment_softmax(dev):
    from ragraph_amd import kernels as K

    rng = _rng(3)
    rowptr, col, val = _rand_csr(rng, 500, 500, 8, empty_rows=False)
    got = K.csr_row_normalize(_t(rowptr, dev), _t(val, dev)).cpu().numpy()
    assert np.array_equal(got, cref.csr_row_normalize(rowptr, val))
    rowptr2, _, x = _rand_csr(rng, 400, 400, 5, empty_rows=True)
    got = K.segment_softmax(_t(rowptr2, dev), _t(x, dev)).cpu().numpy()
    ref = cref.segment_softmax(rowptr2, x)
    assert np.allclose(got, ref, rtol=1e-6, atol=1e-7)
    sums = np.add.reduceat(got, rowptr2[:-1][np.diff(rowptr2) > 0])
    assert np.allclose(sums, 1.0, atol=1e-5)


@pytest.mark.parametrize("Kd,n,Nout,act", [(128, 5000, 256, 2), (64, 4097, 256, 1), (128, 300, 70, 0), (64, 1, 3, 3), (128, 777, 300, 2)])
def test_spmm_linear_one_launch_bit_exact(dev, Kd, n, Nout, act):
    """The aggregate-first GCN layer in one launch (ragraph_spmm_linear_f32) = ragraph_spmm_csr_f32 followed by
    ragraph_linear_f32, bit for bit, and = the oracle's two steps: empty rows, a long row, a hub row past the 4096-edge block,
    ragged last stage, an output wider than one 256-column block, a slice of the row pointers."""
    from ragraph_amd import kernels as K

    rng = _rng(Kd + n + Nout)
    ncols = max(n, 50)
    rowptr, col, val = _rand_csr(rng, n, ncols, 7)
    if n > 100:   # a hub row: 2 blocks + a tail
        deg = np.diff(rowptr)
        deg[50] = 2 * 4096 + 19
        rowptr = np.zeros(n + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum(deg)
        col = rng.integers(0, ncols, rowptr[-1]).astype(np.int32)
        val = (rng.random(rowptr[-1], dtype=np.float32) - 0.3)
    X = rng.standard_normal((ncols, Kd), dtype=np.float32)
    W = (rng.standard_normal((Nout, Kd), dtype=np.float32) * 0.1).astype(np.float32)
    b = rng.standard_normal(Nout, dtype=np.float32)
    args = (_t(rowptr, dev), _t(col, dev), _t(val, dev), _t(X, dev))
    for bias in (b, None):
        bd = None if bias is None else _t(bias, dev)
        got = K.spmm_linear(*args, _t(W, dev), bd, act=act, alpha=0.25).cpu().numpy()
        two = K.linear(K.spmm_csr(*args, long_rows=False), _t(W, dev), bd, act=act, alpha=0.25).cpu().numpy()
        assert np.array_equal(got, two), "one launch != two launches"
        ref = cref.linear(cref.spmm_csr(rowptr, col, val, X), W, bias=bias, act=act, alpha=0.25)
        assert np.array_equal(got, ref), "one launch != oracle"
    if n > 100:   # rows [lo, hi) through a slice of the row pointers (what a query-sharded rank encodes first)
        lo, hi = 37, n - 11
        got = K.spmm_linear(_t(rowptr, dev)[lo:hi + 1], *args[1:], _t(W, dev), _t(b, dev), act=act, alpha=0.25).cpu().numpy()
        assert np.array_equal(got, ref_rows(rowptr, col, val, X, W, b, act, lo, hi))


def ref_rows(rowptr, col, val, X, W, b, act, lo, hi):
    return cref.linear(cref.spmm_csr(rowptr, col, val, X), W, bias=b, act=act, alpha=0.25)[lo:hi]


@pytest.mark.parametrize("D", [64, 256, 20])
def test_spmm_and_softmax_hub_rows_bit_exact(dev, D):
    """Power-law shape: a few rows hold thousands of edges (one of them most of the graph).  Rows longer than 4096 edges
    are summed in blocks of 4096 (block chains from +0, added in block order) -- by the row's own lanes without a
    workspace, by the whole chip with one; both must give the oracle's bits, and short rows stay single chains."""
    from ragraph_amd import kernels as K
    from ragraph_amd.graph import CSRGraph

    rng = _rng(500 + D)
    n, ncols = 300, 5000
    deg = rng.integers(0, 12, n)
    deg[7], deg[8], deg[150], deg[299] = 4096, 4097, 3 * 4096 + 5, 40_000   # at, just past and far past the block size
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    nnz = int(rowptr[-1])
    col = rng.integers(0, ncols, nnz).astype(np.int32)
    val = (rng.random(nnz, dtype=np.float32) - 0.3)
    X = rng.standard_normal((ncols, D), dtype=np.float32)
    b = rng.standard_normal(D, dtype=np.float32)
    Yin = rng.standard_normal((n, D), dtype=np.float32)
    ref = cref.spmm_csr(rowptr, col, val, X, bias=b, act=2, alpha=0.25, beta=0.5, Y_in=Yin)
    args = (_t(rowptr, dev), _t(col, dev), _t(val, dev), _t(X, dev))
    for long_rows in (False, True):
        got = K.spmm_csr(*args, bias=_t(b, dev), act=2, alpha=0.25, beta=0.5, y_in=_t(Yin, dev), long_rows=long_rows)
        assert np.array_equal(got.cpu().numpy(), ref), f"long_rows={long_rows}"
    g = CSRGraph(_t(rowptr, dev), _t(col, dev), _t(val, dev), n)
    assert g.has_long_rows
    if D == 64:
        x = rng.standard_normal(nnz).astype(np.float32)
        sref = cref.segment_softmax(rowptr, x)
        for long_rows in (False, True):
            got = K.segment_softmax(_t(rowptr, dev), _t(x, dev), long_rows=long_rows).cpu().numpy()
            assert np.allclose(got, sref, rtol=1e-6, atol=1e-9), f"long_rows={long_rows}"
            sums = np.add.reduceat(got, rowptr[:-1][np.diff(rowptr) > 0])
            assert np.allclose(sums, 1.0, atol=1e-4)
        a = K.segment_softmax(_t(rowptr, dev), _t(x, dev), long_rows=False)
        c = K.segment_softmax(_t(rowptr, dev), _t(x, dev), long_rows=True)
        assert torch.equal(a, c)   # the two paths agree bit for bit


def test_segment_reduce_axpby_softmax_proto(dev):
    from ragraph_amd import kernels as K

    rng = _rng(21)
    X = rng.standard_normal((300, 256), dtype=np.float32)
    sizes = np.array([10, 1, 80, 0, 39, 170])
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    w = rng.standard_normal(256, dtype=np.float32)
    for ww, mm in [(None, True), (w, False), (None, False)]:
        got = K.segment_reduce(_t(X, dev), _t(seg, dev), None if ww is None else _t(ww, dev), mean_mode=mm)
        ref = cref.segment_reduce(X, seg, ww, mean_mode=mm)
        ok = np.isfinite(ref)
        assert np.array_equal(got.cpu().numpy()[ok], ref[ok])
    a = rng.standard_normal((100, 256), dtype=np.float32)
    b = rng.standard_normal((100, 256), dtype=np.float32)
    assert np.array_equal(K.axpby(_t(a, dev), 0.7, _t(b, dev), 0.3).cpu().numpy(), cref.axpby(a, 0.7, b, 0.3))
    lg = rng.standard_normal((1000, 3), dtype=np.float32) * 3
    rl = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 1000)]
    got = K.softmax_mix(_t(lg, dev), _t(rl, dev), 0.5).cpu().numpy()
    assert np.allclose(got, cref.softmax_mix(lg, rl, 0.5), atol=1e-6)
    got = K.softmax_mix(_t(lg, dev), None, 0.0, log_mode=True).cpu().numpy()
    assert np.allclose(got, cref.softmax_mix(lg, None, 0.0, log_mode=True), atol=1e-5)
    emb = rng.standard_normal((50, 256), dtype=np.float32)
    for C in (2, 6):
        proto = rng.standard_normal((C, 256), dtype=np.float32)
        for mode in (0, 1, 2):
            got = K.proto_cosine(_t(emb, dev), _t(proto, dev), mode).cpu().numpy()
            assert np.allclose(got, cref.proto_cosine(emb, proto, mode), atol=1e-5)


@pytest.mark.parametrize("n,D,H,C", [(1, 128, 128, 7), (16, 256, 256, 2), (2708, 128, 128, 7), (333, 96, 50, 3),
                                      (1000, 130, 67, 5)])
def test_fuse_decode_same_bits_as_separate_entries(dev, n, D, H, C):
    """K5 + K6 in one launch (RAGraph.py:53-57 + TaskDecoder.py:14-17): same bits as axpby -> linear(LeakyReLU) -> linear
    -> softmax_mix on the HIP path, logits chain bit-identical to the oracle's (softmax within 1e-6: expf)."""
    from ragraph_amd import kernels as K

    rng = _rng(n + D + H + C)
    q = rng.standard_normal((n, D), dtype=np.float32)
    r = rng.standard_normal((n, D), dtype=np.float32)
    W1 = (rng.standard_normal((H, D), dtype=np.float32) / np.sqrt(D)).astype(np.float32)
    b1 = rng.standard_normal(H, dtype=np.float32)
    W2 = (rng.standard_normal((C, H), dtype=np.float32) / np.sqrt(H)).astype(np.float32)
    b2 = rng.standard_normal(C, dtype=np.float32)
    rl = np.eye(C, dtype=np.float32)[rng.integers(0, C, n)]
    d = lambda a: _t(a, dev)
    got = K.fuse_decode(d(q), d(r), 0.7, 0.3, d(W1), d(b1), 0.01, d(W2), d(b2), d(rl), 0.4)
    hid = K.axpby(d(q), 0.7, d(r), 0.3)
    sep = K.softmax_mix(K.linear(K.linear(hid, d(W1), d(b1), act=K.ACT_LEAKY, alpha=0.01), d(W2), d(b2)), d(rl), 0.4)
    assert torch.equal(got, sep)
    ref = cref.softmax_mix(cref.linear(cref.linear(cref.axpby(q, 0.7, r, 0.3), W1, b1, act=cref.ACT_LEAKY, alpha=0.01),
                                       W2, b2), rl, 0.4)
    assert np.allclose(got.cpu().numpy(), ref, atol=1e-6)
    # no biases, no label term
    got = K.fuse_decode(d(q), d(r), 0.5, 0.5, d(W1), None, 0.01, d(W2), None, None, 0.0)
    sep = K.softmax_mix(K.linear(K.linear(K.axpby(d(q), 0.5, d(r), 0.5), d(W1), act=K.ACT_LEAKY, alpha=0.01), d(W2)), None, 0.0)
    assert torch.equal(got, sep)


def test_errors_are_loud(dev):
    from ragraph_amd import kernels as K

    with pytest.raises(K.RagraphNativeError):
        K.topk_cosine(torch.randn(4, 100, device=dev), torch.randn(50, 64, device=dev), 5)  # widths differ
    with pytest.raises(K.RagraphNativeError):
        K.topk_cosine(torch.randn(4, 64, device=dev), torch.randn(3, 64, device=dev), 5)  # k > N
    with pytest.raises(K.RagraphNativeError):
        K.topk_cosine(torch.randn(4, 64), torch.randn(30, 64), 5)  # CPU tensors: no fallback


@pytest.mark.parametrize(
    "B,N,k",
    [
        (129, 31, 5),          # one ragged stage
        (200, 64, 10),         # exactly two stages: prologue only
        (257, 97, 3),          # three stages + ragged tail, two query tiles
        (300, 4099, 10),       # several ring generations
        (513, 70001, 14),      # several splits; largest k the 4-slot ring holds
        (1000, 20000, 1),
    ],
)
def test_topk_cosine_packed_bank_bit_exact(dev, topk_family, B, N, k):
    """LDS-DMA ring over the packed bank copy (D = 256, B > 128, k <= 14) vs the oracle, and the pack layout itself."""
    from ragraph_amd import kernels as K

    rng = _rng(B + 3 * N + k)
    kn = _bank(rng, N, 256)
    if N > 8:
        kn[N // 2:] = kn[: N - N // 2]                    # duplicates -> exact ties across stages
    q = rng.standard_normal((B, 256), dtype=np.float32)
    q[B // 2] = 0.0
    knd = _t(kn, dev)
    kp = K.pack_keys(knd)
    want = np.concatenate([kn[:, 0::2], kn[:, 1::2]], axis=1)
    assert np.array_equal(kp.cpu().numpy(), want)
    assert K.packed_keys_help(B, 256, k)
    s, i = K.topk_cosine(_t(q, dev), knd, k, idx_base=11, keys_packed=kp)
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=11)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(s.cpu().numpy(), rs)
    # shapes outside the DMA path ignore the packed copy (same entry point, same bits)
    s2, i2 = K.topk_cosine(_t(q[:7], dev), knd, min(k, 3), keys_packed=kp)
    rs2, ri2 = cref.topk_cosine(q[:7], kn, min(k, 3))
    assert np.array_equal(i2.cpu().numpy(), ri2) and np.array_equal(s2.cpu().numpy(), rs2)


@pytest.mark.parametrize("D", [64, 128])
def test_pack_keys_other_dims(dev, D):
    from ragraph_amd import kernels as K

    rng = _rng(D)
    kn = _bank(rng, 37, D)
    kp = K.pack_keys(_t(kn, dev)).cpu().numpy()
    assert np.array_equal(kp, np.concatenate([kn[:, 0::2], kn[:, 1::2]], axis=1))


def test_topk_cosine_fuzz_against_oracle(dev, topk_family):
    """60 random shapes across all kernel paths (streaming with 1-8 groups, with / without the pre-pass; tile kernel
    with 1-3 query tiles, ring and barrier variants; materialised k > 32), ragged sizes, duplicate keys, zero queries."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(2024)
    for trial in range(60):
        D = int(rng.choice([64, 128, 256]))
        B = int(rng.choice([1, 2, 3, 5, 15, 16, 17, 31, 33, 64, 100, 128, 129, 200, 257, 520]))
        N = int(rng.choice([1, 2, 7, 31, 33, 100, 511, 1000, 4097, 9999, 70001]))
        k = int(min(N, rng.choice([1, 2, 3, 5, 10, 17, 31, 32, 33, 50, 64])))
        keys = rng.standard_normal((N, D), dtype=np.float32)
        if trial % 3 == 0 and N > 4:
            keys[N // 2:] = keys[: N - N // 2]            # duplicates -> exact ties
        kn = cref.normalize_rows(keys)
        q = rng.standard_normal((B, D), dtype=np.float32)
        if trial % 5 == 0:
            q[rng.integers(0, B)] = 0.0                   # zero-norm query
        base = int(rng.choice([0, 7, 1_000_000]))
        s, i = K.topk_cosine(_t(q, dev), _t(kn, dev), k, idx_base=base)
        rs, ri = cref.topk_cosine(q, kn, k, idx_base=base)
        assert np.array_equal(i.cpu().numpy(), ri), f"indices differ: B={B} N={N} D={D} k={k}"
        assert np.array_equal(s.cpu().numpy(), rs), f"scores differ: B={B} N={N} D={D} k={k}"


def test_spmm_linear_fuzz_against_oracle(dev):
    """Random shapes for the CSR SpMM (all lane-group variants, empty / long rows, every epilogue) and the dense kernel."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(77)
    for trial in range(40):
        n = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1025]))
        ncols = int(rng.choice([1, 7, 64, 500]))
        D = int(rng.choice([4, 8, 12, 60, 64, 68, 128, 192, 256, 260, 512]))
        rowptr, col, val = _rand_csr(rng, n, ncols, int(rng.choice([0, 1, 4, 20])), empty_rows=True)
        X = rng.standard_normal((ncols, D), dtype=np.float32)
        act = int(rng.integers(0, 5))
        b = rng.standard_normal(D, dtype=np.float32) if trial % 2 else None
        Yin = rng.standard_normal((n, D), dtype=np.float32) if trial % 3 == 0 else None
        got = K.spmm_csr(_t(rowptr, dev), _t(col, dev), _t(val, dev), _t(X, dev), bias=None if b is None else _t(b, dev),
                         act=act, alpha=0.3, beta=0.5, y_in=None if Yin is None else _t(Yin, dev)).cpu().numpy()
        ref = cref.spmm_csr(rowptr, col, val, X, bias=b, act=act, alpha=0.3, beta=0.5, Y_in=Yin)
        if act == 4:  # ELU calls expm1f: libm differs between host and device
            assert np.allclose(got, ref, rtol=1e-6, atol=1e-6), f"spmm n={n} D={D} act={act}"
        else:
            assert np.array_equal(got, ref), f"spmm n={n} D={D} act={act}"
        M, Kd, Nd = (int(rng.choice([1, 31, 64, 65, 200, 333])), int(rng.choice([1, 5, 32, 33, 100, 257, 64, 132])),
                     int(rng.choice([1, 3, 64, 70, 128, 200])))
        A = rng.standard_normal((M, Kd), dtype=np.float32)
        W = rng.standard_normal((Nd, Kd), dtype=np.float32)
        assert np.array_equal(K.linear(_t(A, dev), _t(W, dev)).cpu().numpy(), cref.linear(A, W)), f"linear {M}x{Kd}x{Nd}"


@pytest.mark.parametrize(
    "B,N,k",
    [
        (700, 40000, 10),      # sample = 16384 keys, two query tiles (one ragged), one group of workgroups
        (513, 33000, 1),
        (1500, 70001, 14),     # N not a multiple of 32: zero-padded bf16 rows are masked by index
        (600, 20000, 32),
    ],
)
def test_topk_cosine_filtered_bit_exact(dev, B, N, k):
    """bf16-filtered exact top-k vs the oracle: indices and scores bit-identical, no overflow on random banks."""
    from ragraph_amd import kernels as K

    rng = _rng(B + N + k)
    kn = _bank(rng, N, 256)
    q = rng.standard_normal((B, 256), dtype=np.float32)
    knd = _t(kn, dev)
    kb = K.keys_to_bf16(knd)
    npad = -(-N // 256) * 256   # bf16 copy padded to whole stages + its error row, then the int8 copy (half the rows) + its row
    # (+ the int8 copy's granule table, + slack the int8 levels' last stage may read)
    assert kb.shape[0] == npad + 1 + npad // 2 + 1 + (npad // 4096 + 8) + 256
    s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, kb, k, idx_base=9)
    assert over == 0
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=9)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(s.cpu().numpy(), rs)


@pytest.mark.parametrize("D,B,N,k", [(64, 700, 40000, 10), (64, 1500, 70001, 5), (128, 513, 33000, 7), (128, 900, 66000, 32)])
def test_topk_cosine_filtered_other_dims_bit_exact(dev, D, B, N, k):
    """The filter at D = 64 (the edge flavour's embedding size) and 128: 8 / 4 sub-tiles per ring stage, other swizzles."""
    from ragraph_amd import kernels as K

    rng = _rng(D + B + N + k)
    kn = _bank(rng, N, D)
    kn[N // 2:N // 2 + 40] = kn[:40]                      # exact duplicates -> ties
    q = rng.standard_normal((B, D), dtype=np.float32)
    knd = _t(kn, dev)
    s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), k, idx_base=2)
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=2)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(s.cpu().numpy(), rs)


def test_topk_cosine_filtered_overflow_falls_back(dev):
    """A bank of near-duplicates puts thousands of keys within EPS of every query's k-th best, and a zero query ties
    every key at 0: the final rescoring level lists those queries and the call's last launch recomputes them with an
    exact fp32 scan ON THE DEVICE (no host read-back) -- the result is still bit-identical to the oracle."""
    from ragraph_amd import kernels as K

    rng = _rng(5)
    base = rng.standard_normal((1, 256), dtype=np.float32)
    keys = base + 1e-3 * rng.standard_normal((20000, 256), dtype=np.float32)
    kn = cref.normalize_rows(keys)
    q = np.concatenate([base + 1e-3 * rng.standard_normal((300, 256), dtype=np.float32),
                        np.zeros((1, 256), dtype=np.float32),
                        rng.standard_normal((50, 256), dtype=np.float32)]).astype(np.float32)
    knd = _t(kn, dev)
    s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), 10)
    assert over >= 301
    rs, ri = cref.topk_cosine(q, kn, 10)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(s.cpu().numpy(), rs)
    # the bank-side dispatch notices that this bank defeats the filter and keeps it on the fp32 kernels afterwards
    # (80000 DISTINCT near-duplicates: large enough for the filtered path; exact copies would be collapsed by KeyIndex,
    # tests/test_gpu_dedup.py)
    kn4 = cref.normalize_rows(base + 1e-3 * rng.standard_normal((80000, 256), dtype=np.float32))
    big = _t(kn4, dev)
    index = K.KeyIndex(big)
    qd = _t(q, dev)
    s1, i1 = index.topk(qd, 10)
    assert index._collapsed is False
    assert index._bf16 is not None and not index._filter_off   # filtered call; its overflow count is not read back ...
    torch.cuda.synchronize()
    s2, i2 = index.topk(qd, 10)                                 # ... it arrives later and is noticed at the next call
    assert index._filter_off and index.overflowed_queries >= 301
    assert torch.equal(i1, i2) and torch.equal(s1, s2)
    rs4, ri4 = cref.topk_cosine(q, kn4, 10)
    assert np.array_equal(i1.cpu().numpy(), ri4) and np.array_equal(s1.cpu().numpy(), rs4)


def test_topk_cosine_filtered_overflow_mid_batch_scans_in_the_rescoring_wave(dev):
    """2048..16384 queries (one wave per query in the rescoring kernel): an overflowed query is answered by that wave's own
    exact scan on the final level -- no fallback launch -- still the oracle's bits; ordinary queries beside it unaffected."""
    from ragraph_amd import kernels as K

    rng = _rng(23)
    base = rng.standard_normal((1, 128), dtype=np.float32)
    kn = cref.normalize_rows(np.concatenate([base + 1e-3 * rng.standard_normal((9000, 128), dtype=np.float32),
                                             rng.standard_normal((3000, 128), dtype=np.float32)]))
    q = rng.standard_normal((2600, 128), dtype=np.float32)
    q[5] = base[0]
    q[77] = 0.0
    q[2599] = base[0] + 1e-3 * rng.standard_normal(128, dtype=np.float32)
    knd = _t(kn, dev)
    s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), 10, idx_base=3)
    assert int(over) >= 3
    rs, ri = cref.topk_cosine(q, kn, 10, idx_base=3)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)


def test_filtered_bound_pass_shapes_match_fp32_kernels(dev):
    """Banks of >= 65536 keys take their first bound from the bound pass (k group maxima of approximate scores over a
    prefix).  Shapes around its edges -- the smallest such bank, one and 32 groups, every D, one / two query groups per
    wave, the slab-sized and the large-batch schedules, a prefix clipped by the first level -- against the fp32 kernels
    (themselves checked against the oracle above): same bits, no overflow on ordinary data."""
    from ragraph_amd import _native as Nn
    from ragraph_amd import kernels as K
    import ctypes

    g = torch.Generator(device=dev).manual_seed(77)
    plan = (ctypes.c_int64 * 7)()
    seen_modes = set()
    for (B, N, D, k) in [(13, 65536, 64, 32), (256, 65536, 256, 1), (300, 70003, 128, 10), (513, 140001, 256, 17),
                         (1500, 300000, 64, 32), (5000, 200000, 256, 10), (20000, 150000, 128, 5),
                         (17000, 65536, 64, 32), (40, 1_000_000, 256, 32)]:
        Nn.lib().ragraph_topk_cosine_filtered_plan(B, N, D, k, plan)
        seen_modes.add(int(plan[1]))
        kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
        q = torch.randn(B, D, device=dev, generator=g)
        q[0] = kn[N - 1]  # a winner in the bank's last row
        s1, i1, over = K.topk_cosine_filtered(q, kn, K.keys_to_bf16(kn), k, idx_base=9)
        s0, i0 = K.topk_cosine(q, kn, k, idx_base=9)
        assert over == 0, (B, N, D, k)
        assert torch.equal(i0, i1) and torch.equal(s0, s1), (B, N, D, k, list(plan))
        assert int(i1[0, 0]) == N - 1 + 9
    assert 2 in seen_modes


def test_key_index_dispatch_same_bits(dev, monkeypatch):
    """KeyIndex picks the kernel by shape (streaming, tile + packed copy, bf16-filtered with one or two query groups per
    wave and the slab or tile-kernel level 0): every choice returns the bits of the oracle, and RAGRAPH_EXACT_FP32=1
    keeps a call on the fp32 kernels."""
    from ragraph_amd import kernels as K

    rng = _rng(99)
    kn = _bank(rng, 70000, 256)
    knd = _t(kn, dev)
    index = K.KeyIndex(knd)
    for B in (3, 40, 200, 800):  # >= 65536 keys: filtered at every batch size (direct kernel up to 256, ring kernel beyond)
        q = rng.standard_normal((B, 256), dtype=np.float32)
        assert K.filter_helps(B, 70000, 256, 10)
        s, i = index.topk(_t(q, dev), 10, idx_base=4)
        rs, ri = cref.topk_cosine(q, kn, 10, idx_base=4)
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    assert index._bf16 is not None
    small = K.KeyIndex(_t(kn[:20000], dev))  # a bank below 32768 keys: streaming fp32 kernel, score slab, tile kernel; 600: filtered
    for B in (3, 40, 300, 600):
        q = rng.standard_normal((B, 256), dtype=np.float32)
        assert K.filter_helps(B, 20000, 256, 10) == (B >= 512)
        s, i = small.topk(_t(q, dev), 10, idx_base=4)
        rs, ri = cref.topk_cosine(q, kn[:20000], 10, idx_base=4)
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    # batches beyond MAX_FILTERED_BATCH go through the filtered path in slabs (bounded workspace): same rows, same order
    monkeypatch.setattr(K.KeyIndex, "MAX_FILTERED_BATCH", 300)
    q = rng.standard_normal((800, 256), dtype=np.float32)
    s, i = index.topk(_t(q, dev), 10, idx_base=4)
    rs, ri = cref.topk_cosine(q, kn, 10, idx_base=4)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    monkeypatch.setenv("RAGRAPH_EXACT_FP32", "1")
    assert not K.filter_helps(800, 70000, 256, 10)
    q = rng.standard_normal((200, 256), dtype=np.float32)  # tile kernel with the packed copy
    s, i = index.topk(_t(q, dev), 10, idx_base=4)
    rs, ri = cref.topk_cosine(q, kn, 10, idx_base=4)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    assert index._packed is not None


def test_topk_cosine_filtered_fuzz_against_oracle(dev):
    """Random shapes through the bf16-filtered path (1-3 filter levels, ragged query tiles, N not a multiple of a
    stage, every k up to 32, exact duplicates = ties at the k-th place, zero queries = overflow + fp32 fallback):
    always the oracle's bits."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(4242)
    for trial in range(20):
        B = int(rng.choice([1, 13, 37, 256, 300, 513, 900, 1500]))
        N = int(rng.choice([300, 1000, 4100, 16500, 20001, 40000, 70003, 140001]))
        k = int(rng.choice([1, 2, 5, 10, 17, 32]))
        D = int(rng.choice([64, 128, 256, 256]))
        keys = rng.standard_normal((N, D), dtype=np.float32)
        if trial % 3 == 0 and N > 200:
            keys[N // 2:N // 2 + 64] = keys[:64]          # exact duplicates -> ties
        kn = cref.normalize_rows(keys)
        q = rng.standard_normal((B, D), dtype=np.float32)
        if trial % 4 == 1:
            q[rng.integers(0, B)] = 0.0                   # zero query: every key ties at 0 -> overflow -> fallback
        base = int(rng.choice([0, 11]))
        knd = _t(kn, dev)
        s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), k, idx_base=base)
        rs, ri = cref.topk_cosine(q, kn, k, idx_base=base)
        assert np.array_equal(i.cpu().numpy(), ri), f"indices differ: B={B} N={N} D={D} k={k} (overflowed {over})"
        assert np.array_equal(s.cpu().numpy(), rs), f"scores differ: B={B} N={N} D={D} k={k}"


def test_topk_cosine_filtered_adversarial_rounding(dev):
    """Inputs built to make the bf16 rounding errors of a query and its best key ALIGN (half of the elements sit just
    below a rounding midpoint, so bf16 rounds them all down): the winner's approximate score drops by ~0.4 % while the
    exact threshold, set by near-duplicates in the first level, stays ~0.05 % below it.  The filter's bound must keep
    the winner; the result must still be the oracle's bits."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(31337)
    N, D, k, nq = 70000, 256, 5, 64
    keys = rng.standard_normal((N, D)).astype(np.float32)
    keys /= np.linalg.norm(keys, axis=1, keepdims=True)
    qs = []
    for j in range(nq):
        pick = rng.permutation(D) < D // 2
        sign = rng.choice([-1.0, 1.0], D)
        a = 2.0 ** -4 * (1 + 2.0 ** -8 * 0.97)                 # rounds DOWN to 2^-4 in bf16 (0.97 of half an ulp)
        b = np.sqrt((1.0 - (D // 2) * a * a) / (D - D // 2))  # unit norm
        v = (np.where(pick, a, b) * sign).astype(np.float32)
        qs.append(v)
        keys[N - 1 - j] = v                                    # the true best key, met in the LAST filter level
        for d in range(8):                                     # near-duplicates in the first 4096 rows set the bound
            noise = rng.standard_normal(D).astype(np.float32)
            w = v + (0.02 + 0.004 * d) * noise / np.linalg.norm(noise)
            keys[j * 8 + d] = w / np.linalg.norm(w)
    q = np.stack(qs).astype(np.float32)
    kn = cref.normalize_rows(keys)
    knd = _t(kn, dev)
    s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), k)
    rs, ri = cref.topk_cosine(q, kn, k)
    assert np.array_equal(ri[:, 0], N - 1 - np.arange(nq))     # the construction: each query's best key is its twin
    # how much the bf16 score of the twin is off (the thing the bound has to cover)
    import torch
    qn = cref.normalize_rows(q)
    qb = torch.from_numpy(qn).to(torch.bfloat16).float().numpy()
    kb = torch.from_numpy(kn[ri[:, 0]]).to(torch.bfloat16).float().numpy()
    drop = rs[:, 0] - (qb * kb).sum(1)
    assert drop.min() > 0.002, drop.min()                      # aligned errors: an order of magnitude above random
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)


@pytest.mark.parametrize("D,B,N,k", [(128, 2708, 10000, 5), (256, 33, 1113, 3), (64, 1, 640, 1), (128, 100, 5003, 8),
                                     (256, 300, 4100, 16), (64, 777, 20000, 10), (256, 64, 2048, 4)])
def test_topk_cosine_fused_bit_exact(dev, D, B, N, k):
    """The single-launch small-bank kernel (csrc/topk_fused.hip: normalise, bf16 bound, bf16 filter, exact rescoring and
    canonical selection in one workgroup per 32 queries) against the oracle: indices and score bits, ragged tiles and
    banks, exact duplicate keys (ties at the k-th place), an index base."""
    from ragraph_amd import kernels as K

    rng = _rng(D + B + N + k)
    kn = _bank(rng, N, D)
    kn[N // 2:N // 2 + 40] = kn[:40]                      # exact duplicates -> ties, canonical order decides
    q = rng.standard_normal((B, D), dtype=np.float32)
    q[B // 2] = 3.0 * kn[7]                               # a query that IS a stored key (score 1, two copies)
    knd = _t(kn, dev)
    s, i = K.topk_cosine_fused(_t(q, dev), knd, K.keys_to_bf16(knd), k, idx_base=11)
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=11)
    assert np.array_equal(i.cpu().numpy(), ri)
    assert np.array_equal(s.cpu().numpy(), rs)


def test_topk_cosine_fused_overflow_zero_queries_and_dispatch(dev, monkeypatch):
    """Near-duplicate bank: thousands of keys within eps of the k-th best overflow the 512-slot candidate list, a zero
    query ties every key at 0 -- the query's wave answers with an exact scan, still the oracle's bits.  And KeyIndex sends
    a many-queries-small-bank call to this kernel (and nothing else), same bits as the fp32 kernels."""
    from ragraph_amd import kernels as K

    rng = _rng(17)
    base = rng.standard_normal((1, 128), dtype=np.float32)
    kn = cref.normalize_rows(base + 1e-3 * rng.standard_normal((6000, 128), dtype=np.float32))
    q = np.concatenate([base + 1e-3 * rng.standard_normal((40, 128), dtype=np.float32), np.zeros((1, 128), dtype=np.float32),
                        rng.standard_normal((30, 128), dtype=np.float32)]).astype(np.float32)
    knd = _t(kn, dev)
    s, i = K.topk_cosine_fused(_t(q, dev), knd, K.keys_to_bf16(knd), 10)
    rs, ri = cref.topk_cosine(q, kn, 10)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    # dispatch: 2708 x 10000 x 64 (a Cora-sized batch against a 10k bank at the edge flavour's width)
    kn2 = _bank(rng, 10000, 64)
    q2 = rng.standard_normal((2708, 64), dtype=np.float32)
    index = K.KeyIndex(_t(kn2, dev))
    calls = []
    real = K.topk_cosine_fused
    monkeypatch.setattr(K, "topk_cosine_fused", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    assert K.fused_helps(2708, 10000, 64, 5) and not K.fused_helps(2708, 1_000_000, 256, 10)
    assert not K.fused_helps(2708, 10000, 128, 5) and not K.fused_helps(256, 2000, 64, 5)
    s2, i2 = index.topk(_t(q2, dev), 5)
    assert calls == [1]
    s3, i3 = K.topk_cosine(_t(q2, dev), _t(kn2, dev), 5)
    assert torch.equal(i2, i3) and torch.equal(s2, s3)
    rs2, ri2 = cref.topk_cosine(q2, kn2, 5)
    assert np.array_equal(i2.cpu().numpy(), ri2) and np.array_equal(s2.cpu().numpy(), rs2)


@pytest.mark.parametrize("D,B,N,k,levels", [(256, 700, 70000, 10, 3), (128, 513, 33000, 7, 3), (256, 3000, 40000, 5, 1),
                                            (64, 900, 70000, 10, 3), (64, 3000, 100000, 5, 2),
                                            (128, 17000, 70000, 10, -1), (256, 17000, 66000, 32, -1), (64, 20000, 131072, 10, -1)])
def test_topk_cosine_filtered_int8_levels_bit_exact(dev, monkeypatch, D, B, N, k, levels):
    """Filter levels on the INT8 copy (v_mfma_i32_16x16x64_i8, integer thresholds; csrc/filter_common.h): forced on every
    level of small shapes (RAGRAPH_FILTER_I8), and the product rule -- whatever levels the schedule plans for the shape --
    as it stands (levels = -1).  Always the oracle's bits: exact duplicates, a query that is a stored key, a zero query
    (scale 0: everything passes, the exact scan answers), ragged tiles and banks."""
    from ragraph_amd import kernels as K

    if levels >= 0:
        monkeypatch.setenv("RAGRAPH_FILTER_I8", str(levels))
    else:
        monkeypatch.delenv("RAGRAPH_FILTER_I8", raising=False)
    rng = _rng(D + B + N + k)
    kn = _bank(rng, N, D)
    kn[N // 2:N // 2 + 40] = kn[:40]
    q = rng.standard_normal((B, D), dtype=np.float32)
    q[3] = 2.5 * kn[11]
    q[B - 1] = 0.0
    knd = _t(kn, dev)
    s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), k, idx_base=5)
    assert int(over) <= 1                                   # (only the zero query may go to a scan: the FILTERED path produced every other row)
    rows = np.arange(B) if B <= 4000 else np.unique(np.concatenate([[3, B - 1], rng.integers(0, B, 700)]))
    rs, ri = cref.topk_cosine(q[rows], kn, k, idx_base=5)
    assert np.array_equal(i.cpu().numpy()[rows], ri)
    assert np.array_equal(s.cpu().numpy()[rows], rs)
    if B > 4000:                                            # every row against the fp32 kernel
        s32, i32 = K.topk_cosine(_t(q, dev), knd, k, idx_base=5)
        assert torch.equal(i, i32) and torch.equal(s, s32)


@pytest.mark.parametrize("D,B,N,k", [(256, 2500, 70000, 10), (128, 2100, 66000, 32), (64, 2048, 70000, 1),
                                     (256, 4100, 131072, 5), (256, 700, 300000, 10), (256, 200, 150000, 10),
                                     (128, 90, 70000, 5), (256, 65, 1000000, 16)])
def test_topk_cosine_filtered_scored_lists_bit_exact(dev, monkeypatch, D, B, N, k):
    """Scored candidate lists (int8 levels of calls of 65 queries and more -- the ring kernel's and the direct kernel's; the
    product rule takes them at D = 256 and k <= 16, forced here at every width): entries {key, I}, rescoring in two rounds -- the 16 largest I first, then only the entries whose I can still
    reach the k-th best found.  Whatever round 1 picks the result is the oracle's: clusters of near-duplicates of 60 / 160 /
    400 keys put a query's list in every branch (most of round 2 beating round 1's k-th pair; more than 64 doing so: the
    plain path; lists beyond 256 entries), exact duplicates tie on I and on the score, a zero query overflows.  With and
    without the scores: the same bits."""
    from ragraph_amd import kernels as K

    monkeypatch.setenv("RAGRAPH_FILTER_I8", "3")
    rng = _rng(3 * D + B + N + k)
    kn = _bank(rng, N, D)
    centres = rng.standard_normal((3, D), dtype=np.float32)
    at = [N // 3, N // 2 + 1000, N - 5000]
    for c, (lo, n) in enumerate(zip(at, (60, 160, 400))):
        kn[lo:lo + n] = cref.normalize_rows(centres[c] + 2e-3 * rng.standard_normal((n, D), dtype=np.float32))
    kn[N // 2:N // 2 + 40] = kn[:40]
    q = rng.standard_normal((B, D), dtype=np.float32)
    for c in range(3):
        q[10 + 4 * c] = centres[c]
        q[11 + 4 * c] = centres[c] + 1e-3 * rng.standard_normal(D, dtype=np.float32)
        q[12 + 4 * c] = 3.0 * kn[at[c] + 7]
    q[3] = 2.5 * kn[11]
    q[B - 1] = 0.0
    knd, qd = _t(kn, dev), _t(q, dev)
    kb = K.keys_to_bf16(knd)
    outs = {}
    for scored in ("1", "0"):
        monkeypatch.setenv("RAGRAPH_FILTER_SCORED", scored)
        s, i, over = K.topk_cosine_filtered(qd, knd, kb, k, idx_base=9)
        assert int(over) <= 1                                   # (only the zero query may go to a scan: the FILTERED path produced every other row)
        outs[scored] = (s, i)
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
    rows = np.unique(np.concatenate([np.arange(24), [B - 1], rng.integers(0, B, 400)]))
    rs, ri = cref.topk_cosine(q[rows], kn, k, idx_base=9)
    assert np.array_equal(outs["1"][1].cpu().numpy()[rows], ri)
    assert np.array_equal(outs["1"][0].cpu().numpy()[rows], rs)
    s32, i32 = K.topk_cosine(qd, knd, k, idx_base=9)            # every row against the fp32 kernel
    assert torch.equal(outs["1"][1], i32) and torch.equal(outs["1"][0], s32)


@pytest.mark.parametrize("B", [1, 40, 300, 2500])
def test_zero_queries_are_answered_without_candidates(dev, B):
    """An all-zero query scores +0 against every key, so no bound can exclude anything: the prepare launch flags it, a
    flagged query passes nothing through the filter levels, and the final level's scan path writes its answer (+0, the
    first k indices) without scanning."""
    from ragraph_amd import kernels as K

    rng = _rng(B)
    N, D, k = 200000, 256, 10
    kn = _bank(rng, N, D)
    q = rng.standard_normal((B, D), dtype=np.float32)
    zeros = sorted({0, B // 2, B - 1})
    q[zeros] = 0.0
    knd = _t(kn, dev)
    s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), k, idx_base=3)
    # (the wide rescoring kernels answer them on their scan path and count them; the one-wave kernels do not)
    assert int(over) in (0, len(zeros))
    s, i = s.cpu().numpy(), i.cpu().numpy()
    for z in zeros:
        assert np.array_equal(i[z], np.arange(3, 3 + k)) and np.all(s[z] == 0) and not np.signbit(s[z]).any()
    rows = np.arange(B) if B <= 300 else rng.integers(0, B, 200)
    rs, ri = cref.topk_cosine(q[rows], kn, k, idx_base=3)
    assert np.array_equal(i[rows], ri) and np.array_equal(s[rows], rs)


def test_int8_levels_at_d64_start_at_whole_stages(dev):
    """D = 64: a stage of the int8 copy is 512 keys, level ends used to be multiples of 256 -- a level that began at an odd
    multiple re-read the previous level's last 256 keys, a winner among those was listed twice and the selection (ranks of
    DISTINCT pairs) went wrong for that query (found by tools/soak_filtered.py: 12 of 5056 rows on this very shape).  The
    inner ends are multiples of 512 now; every row equals the fp32 kernel's."""
    from ragraph_amd import kernels as K

    B, N, D, k = 5056, 593347, 64, 5
    plan = (ctypes.c_int64 * 7)()
    for shape in ((B, N), (20000, 131072 + 256), (100000, 4000000)):
        nlev = K.N.lib().ragraph_topk_cosine_filtered_plan(shape[0], shape[1], D, k, plan)
        assert all(int(plan[3 + l]) % 512 == 0 for l in range(nlev - 1))
    g = torch.Generator(device=dev).manual_seed(0)
    kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    q = torch.randn(B, D, device=dev, generator=g)
    assert K.filtered_i8_levels(B, N, D, k) > 0
    s1, i1, over = K.topk_cosine_filtered(q, kn, K.keys_to_bf16(kn), k, idx_base=5)
    s0, i0 = K.topk_cosine(q, kn, k, idx_base=5)
    assert int(over) == 0 and torch.equal(i0, i1) and torch.equal(s0, s1)


def test_int8_copy_scale_and_error_bound(dev):
    """The int8 copy's tail row on a bank of ordinary rows: the scale = cut / 127 with the cut at most the bank's largest |k_i|
    and at least every NORMAL granule's, and max_k |dk|^2 of the rows dequantised on that grid, against numpy.  (Heavy-tailed
    rows get a scale of their own: tests/test_gpu_i8_classes.py.)"""
    from ragraph_amd import kernels as K

    rng = _rng(41)
    N, D = 3000, 256
    kn = _bank(rng, N, D)
    knd = _t(kn, dev)
    kb = K.keys_to_bf16(knd)
    c = K.int8_copy_classes(kb, N)
    assert c["max_abs"] == float(np.abs(kn).max()) and c["cut"] <= c["max_abs"]
    assert np.float32(c["scale"]) == np.float32(c["cut"]) / np.float32(127.0)
    gk = 32768 // D
    cls = np.repeat(np.abs(np.concatenate([kn, np.zeros((-N % gk, D), np.float32)])).reshape(-1, gk * D).max(1) > c["cut"], gk)[:N]
    assert int(cls.sum()) == c["heavy_granules"] * gk or cls[-1]           # (the last granule may be ragged)
    sk = np.float32(c["scale"])
    rows = kn[~cls]
    ki = np.clip(np.rint(rows * (np.float32(1.0) / sk)), -127, 127).astype(np.float32)
    err2 = ((ki * sk - rows).astype(np.float64) ** 2).sum(1).max()
    assert abs(c["err"] ** 2 - err2) <= 1e-4 * err2


def test_key_index_drops_int8_before_the_filter_when_a_bank_overflows(dev):
    """A clustered bank passes the error-row test (its entries are ordinary) but puts thousands of keys within the INT8
    bound of a query's k-th best -- and only hundreds within the bf16 bound.  The first calls overflow on the int8 levels;
    KeyIndex notices (the count arrives asynchronously), keeps this bank's levels on bf16 and the filter ON; every answer
    is the oracle's."""
    from ragraph_amd import kernels as K

    rng = _rng(91)
    N, D, B, k = 70000, 256, 17000, 10     # (a batch whose schedule plans int8 levels on a bank of this size)
    centre = rng.standard_normal((1, D), dtype=np.float32)
    # (noise 0.5: ~90 keys within the bf16 bound of a query's k-th best, ~12 000 within the int8 bound)
    kn = cref.normalize_rows(np.concatenate([centre + 0.5 * rng.standard_normal((30000, D), dtype=np.float32),
                                             rng.standard_normal((N - 30000, D), dtype=np.float32)]))
    q = (centre + 0.5 * rng.standard_normal((B, D), dtype=np.float32)).astype(np.float32)
    idx = K.KeyIndex(_t(kn, dev))
    qd = _t(q, dev)
    rs, ri = cref.topk_cosine(q[:200], kn, k)
    for _ in range(4):
        s, i = idx.topk(qd, k)
        torch.cuda.synchronize()
        assert np.array_equal(i.cpu().numpy()[:200], ri) and np.array_equal(s.cpu().numpy()[:200], rs)
    assert idx._i8_ok is True                    # ordinary entries: the error row does not give the bank away
    assert idx.overflowed_queries > B // 4       # the int8 levels did overflow ...
    assert idx._i8_off and not idx._filter_off   # ... so int8 is what goes; the bf16 filter stays


@pytest.mark.parametrize("D,B,N,k", [(256, 1, 70000, 10), (256, 16, 100000, 5), (128, 33, 70000, 10), (256, 200, 150000, 10),
                                     (128, 256, 66000, 7), (256, 64, 300000, 10),
                                     # D = 64 (round 5; the edge flavour's width): one MFMA per 16-key half, 8 sub-tiles per unit
                                     (64, 1, 70000, 10), (64, 20, 131072, 10), (64, 40, 200000, 10), (64, 256, 300000, 5),
                                     (64, 130, 70001, 10)])
def test_topk_cosine_filtered_int8_direct_kernel_bit_exact(dev, D, B, N, k):
    """Up to 256 queries against a bank of >= 65 536 keys: the direct kernel's pass runs on the int8 copy (half the stream,
    half the matrix work; the schedule planned for ~3x the candidates) -- the oracle's bits, with ties, a zero query and the
    sliced rescoring of a handful of queries."""
    from ragraph_amd import kernels as K

    rng = _rng(3 * D + B + N + k)
    kn = _bank(rng, N, D)
    kn[N // 2:N // 2 + 40] = kn[:40]
    q = rng.standard_normal((B, D), dtype=np.float32)
    if B > 2:
        q[B - 1] = 0.0
    assert K.filtered_i8_levels(B, N, D, k) >= 1
    knd = _t(kn, dev)
    s, i, over = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), k, idx_base=2)
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=2)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    old = K.set_max_i8_levels(0)                               # the same call kept on bf16: the same bits
    try:
        assert K.filtered_i8_levels(B, N, D, k) == 0
        s2, i2, _ = K.topk_cosine_filtered(_t(q, dev), knd, K.keys_to_bf16(knd), k, idx_base=2)
    finally:
        K.set_max_i8_levels(old)
    assert torch.equal(i, i2) and torch.equal(s, s2)


def test_key_index_small_batches_leave_int8_when_the_bank_overflows(dev):
    """Calls of fewer than 64 queries against a clustered bank: the int8 pass overflows the lists of every call (40 queries
    keep two sub-lists of 2048 each; ~12 000 keys lie within the int8 bound); after the first calls KeyIndex keeps the bank on
    bf16 (it used to judge only single calls of >= 64 queries)."""
    from ragraph_amd import kernels as K

    rng = _rng(93)
    N, D, k = 70000, 256, 10
    centre = rng.standard_normal((1, D), dtype=np.float32)
    kn = cref.normalize_rows(np.concatenate([centre + 0.5 * rng.standard_normal((30000, D), dtype=np.float32),
                                             rng.standard_normal((N - 30000, D), dtype=np.float32)]))
    idx = K.KeyIndex(_t(kn, dev))
    for c in range(6):
        q = (centre + 0.5 * rng.standard_normal((40, D), dtype=np.float32)).astype(np.float32)
        s, i = idx.topk(_t(q, dev), k)
        torch.cuda.synchronize()
        rs, ri = cref.topk_cosine(q, kn, k)
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    assert idx._i8_off and not idx._filter_off and idx.overflowed_queries >= 3


def test_candidate_statistics_and_cost_aware_int8_demotion(dev):
    """A filtered call leaves its sampled candidate counts per level at the end of its workspace.  A bank with a cluster of
    a few thousand keys around the queries passes ~1000 candidates per query on its int8 level WITHOUT overflowing (the bf16
    bound passes a handful): nothing fails, the call is merely slower than on bf16 -- KeyIndex reads the counts
    asynchronously and keeps such a bank off int8 within three calls.  An ordinary bank stays on int8.  Every answer is
    the oracle's."""
    from ragraph_amd import kernels as K

    rng = _rng(97)
    N, D, B, k = 70000, 256, 17000, 10     # (a batch whose schedule plans int8 levels on a bank of this size)
    centre = rng.standard_normal((1, D), dtype=np.float32)
    # (the cluster first: the bound pass samples it, so the first bound is the cluster's own k-th best -- ~40 % of its keys lie
    # within the int8 bound of that, a handful within the bf16 bound)
    kn = cref.normalize_rows(np.concatenate([centre + 0.5 * rng.standard_normal((3000, D), dtype=np.float32),
                                             rng.standard_normal((N - 3000, D), dtype=np.float32)]))
    q = (centre + 0.5 * rng.standard_normal((B, D), dtype=np.float32)).astype(np.float32)
    knd, qd = _t(kn, dev), _t(q, dev)
    s, i, over, stats = K.topk_cosine_filtered(qd, knd, K.keys_to_bf16(knd), k, return_stats=True)
    levels = K.filter_stats_levels(stats.cpu().tolist())
    assert not hasattr(K, "last_filter_stats")   # (no process-global view of "the last call")
    assert int(over) == 0 and len(levels) >= 1 and all(c is not None for _, _, c in levels)
    assert sum(keys for _, keys, _ in levels) == N and levels[-1][0] == "int8"
    assert max(c for dt, _, c in levels if dt == "int8") > 400     # the cluster's level
    rs, ri = cref.topk_cosine(q[:200], kn, k)
    assert np.array_equal(i.cpu().numpy()[:200], ri) and np.array_equal(s.cpu().numpy()[:200], rs)
    idx = K.KeyIndex(knd)
    for _ in range(3):
        s2, i2 = idx.topk(qd, k)
        torch.cuda.synchronize()
        assert torch.equal(i2, i) and torch.equal(s2, s)
    idx.topk(qd, k)
    assert idx._i8_off and not idx._filter_off and idx.overflowed_queries == 0
    assert idx.last_i8_candidates > K.KeyIndex.I8_MAX_CANDIDATES_BASE + K.KeyIndex.I8_MAX_CANDIDATES_PER_KEY * N
    plain = K.KeyIndex(_t(_bank(rng, N, D), dev))
    qp = _t(rng.standard_normal((B, D), dtype=np.float32), dev)
    for _ in range(3):
        plain.topk(qp, k)
        torch.cuda.synchronize()
    plain.topk(qp, k)
    assert not plain._i8_off and plain.last_i8_candidates is not None and 100 < plain.last_i8_candidates < 350


@pytest.mark.parametrize("n,D,deg", [(3000, 256, 9), (5001, 128, 5), (777, 64, 12), (40000, 256, 11), (2500, 512, 4)])
def test_spmm_csr_panels_bit_exact(dev, n, D, deg):
    """The column-panel hops (features [D/32][n][32] between the hops of a propagation): row -> panel, panel -> panel and panel ->
    row give the bits of the row kernel and of the oracle in every combination -- with a hub row of more than 4096 edges (summed
    in blocks of 4096, as everywhere), empty rows, and the ReLU epilogue."""
    from ragraph_amd import kernels as K

    rng = _rng(n + D)
    src = rng.integers(0, n, n * deg)
    dst = rng.integers(0, n, n * deg)
    hub = min(n - 1, 17)
    extra = min(n, 6000)
    src = np.concatenate([src, np.full(extra, hub)])               # row `hub` has > 4096 entries when n allows
    dst = np.concatenate([dst, rng.permutation(n)[:extra]])
    keep = src != 5                                                  # row 5 stays empty
    order = np.lexsort((dst[keep], src[keep]))
    r, c = src[keep][order], dst[keep][order]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr)
    val = rng.standard_normal(r.size).astype(np.float32) * 0.3
    X = rng.standard_normal((n, D), dtype=np.float32)
    rp, cc, vv, Xd = _t(rowptr, dev), _t(c.astype(np.int32), dev), _t(val, dev), _t(X, dev)
    ref = K.spmm_csr(rp, cc, vv, Xd, act=K.ACT_RELU)
    oref = cref.spmm_csr(rowptr, c.astype(np.int32), val, X, act=cref.ACT_RELU)
    assert np.array_equal(ref.cpu().numpy(), oref)
    yp = K.spmm_csr_panels(rp, cc, vv, Xd, x_panels=False, y_panels=True, act=K.ACT_RELU)           # row -> panel
    back = yp.view(D // 32, n, 32).permute(1, 0, 2).reshape(n, D)
    assert torch.equal(back, ref)
    assert torch.equal(K.spmm_csr_panels(rp, cc, vv, Xd, x_panels=False, y_panels=False, act=K.ACT_RELU), ref)   # row -> row, XCD slices
    xp = Xd.view(n, D // 32, 32).permute(1, 0, 2).contiguous().view(n, D)                            # X as panels
    assert torch.equal(K.spmm_csr_panels(rp, cc, vv, xp, x_panels=True, y_panels=False, act=K.ACT_RELU), ref)   # panel -> row
    y2 = K.spmm_csr_panels(rp, cc, vv, xp, x_panels=True, y_panels=True, act=K.ACT_RELU)             # panel -> panel
    assert torch.equal(y2.view(D // 32, n, 32).permute(1, 0, 2).reshape(n, D), ref)
    # two hops through panels = two hops of the row kernel
    two = K.spmm_csr_panels(rp, cc, vv, yp, x_panels=True, y_panels=False, act=K.ACT_RELU)
    assert torch.equal(two, K.spmm_csr(rp, cc, vv, ref, act=K.ACT_RELU))
    # a rank's rows only (key-sharded retrieval: the last hop over a slice of the row pointers, gathering from the whole table)
    lo, hi = n // 3, n // 3 + max(1, n // 5)
    part = K.spmm_csr_panels(rp[lo:hi + 1], cc, vv, yp, x_panels=True, y_panels=False, act=K.ACT_RELU)
    assert part.shape[0] == hi - lo and torch.equal(part, two[lo:hi])


def test_propagation_rows_slice_through_panels(dev):
    """aggregate_k_hop_features(rows=(lo, hi)) at c2's size (the panel hops): the rows of the whole result, bit for bit."""
    from ragraph_amd import data
    from ragraph_amd.ragraph_utils import Propagation

    n, D = 100_000, 256
    from ragraph_amd.graph import CSRGraph
    g = CSRGraph.from_edge_index_sym_normalized(data.synthetic_big_graph(n, 10, seed=5, device=dev), n)
    x = torch.randn(n, D, device=dev)
    whole = Propagation.aggregate_k_hop_features(g, x, 3)
    for lo, hi in ((0, 12_500), (37_500, 50_000), (99_000, 100_000)):
        part = Propagation.aggregate_k_hop_features(g, x, 3, rows=(lo, hi))
        assert part.shape == (hi - lo, D) and torch.equal(part, whole[lo:hi])


def test_propagation_through_the_tiled_hops_opt_in(dev, monkeypatch):
    """RAGRAPH_SPMM_TILED=1: aggregate_k_hop_features on the graph-tiled kernel (row-major in, panel-major between the hops,
    row-major out; the last hop of a rank's row slice on the panel kernel) -- the default path's bits."""
    from ragraph_amd import data
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.ragraph_utils import Propagation

    n, D = 100_000, 256
    g = CSRGraph.from_edge_index_sym_normalized(data.synthetic_big_graph(n, 10, seed=5, device=dev), n)
    x = torch.randn(n, D, device=dev)
    whole = Propagation.aggregate_k_hop_features(g, x, 3)
    assert not g._tile_plans                                  # (the default path made no plan)
    monkeypatch.setenv("RAGRAPH_SPMM_TILED", "1")
    tiled = Propagation.aggregate_k_hop_features(g, x, 3)
    assert g._tile_plans and next(iter(g._tile_plans.values())) is not None
    assert torch.equal(tiled, whole)
    part = Propagation.aggregate_k_hop_features(g, x, 3, rows=(37_500, 50_000))
    assert torch.equal(part, whole[37_500:50_000])
    assert torch.equal(Propagation.aggregate_k_hop_features(g, x, 1), Propagation.aggregate_k_hop_features(g, x, 1))


@pytest.mark.parametrize("n,D,deg", [(3000, 256, 7), (70_001, 128, 9), (70_001, 64, 5), (150_000, 256, 11), (40_000, 512, 6), (129, 256, 3)])
def test_spmm_csr_tiled_bit_exact(dev, n, D, deg):
    """The graph-tiled hop (csrc/sparse.hip spmm_tiled_kernel; CSRGraph.tile_plan): destination chunks whose sums stay in LDS,
    source blocks that fit an L2, edges in (chunk, block, row, column) order -- in every layout combination the bits of the row
    kernel and of the oracle: empty rows at both ends and inside, rows of one edge, duplicate (row, column) entries (their
    stable order decides), several source blocks (a small TILE_SOURCE_BYTES forces them on small graphs), two hops chained."""
    from ragraph_amd import kernels as K
    from ragraph_amd.graph import CSRGraph

    rng = _rng(n + D + deg)
    src = rng.integers(0, n, n * deg)
    dst = rng.integers(0, n, n * deg)
    keep = (src != 0) & (src != n - 1) & (src != n // 2)              # three empty rows
    src, dst = src[keep], dst[keep]
    src = np.concatenate([src, np.full(40, 7 % n)])                   # duplicates of (7, 3): a run of equal columns
    dst = np.concatenate([dst, np.full(40, 3 % n)])
    order = np.lexsort((dst, src))                                    # (stable: columns ascend inside a row)
    r, c = src[order], dst[order]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr)
    val = rng.standard_normal(r.size).astype(np.float32) * 0.3
    X = rng.standard_normal((n, D), dtype=np.float32)
    g = CSRGraph(_t(rowptr, dev), _t(c.astype(np.int32), dev), _t(val, dev), n)
    Xd = _t(X, dev)
    ref = K.spmm_csr(g.rowptr, g.col, g.val, Xd, act=K.ACT_RELU)
    if n <= 70_001:
        assert np.array_equal(ref.cpu().numpy(), cref.spmm_csr(rowptr, c.astype(np.int32), val, X, act=cref.ACT_RELU))
    P = D // 32
    for src_bytes in (CSRGraph.TILE_SOURCE_BYTES, 40_000):           # the product's blocks, and many small ones
        g._tile_plans.clear()
        old, old_near = CSRGraph.TILE_SOURCE_BYTES, CSRGraph.TILE_NEAR_ROWS
        CSRGraph.TILE_SOURCE_BYTES, CSRGraph.TILE_NEAR_ROWS = src_bytes, -1   # (small graphs: every edge is "near" -- judged below)
        try:
            plan = g.tile_plan(P)
        finally:
            CSRGraph.TILE_SOURCE_BYTES, CSRGraph.TILE_NEAR_ROWS = old, old_near
        assert plan is not None and plan.C * plan.RG * 128 >= n and plan.perm.numel() == r.size and int((plan.row3 != -1).sum()) == r.size
        v2 = g.tiled_values(plan, g.val)
        to_rows = lambda y: y.view(P, n, 32).permute(1, 0, 2).reshape(n, D)
        xp = Xd.view(n, P, 32).permute(1, 0, 2).contiguous().view(n, D)
        assert torch.equal(K.spmm_csr_tiled(plan, v2, Xd, n, False, False, act=K.ACT_RELU), ref)             # row -> row
        yp = K.spmm_csr_tiled(plan, v2, Xd, n, False, True, act=K.ACT_RELU)                                 # row -> panel
        assert torch.equal(to_rows(yp), ref)
        assert torch.equal(K.spmm_csr_tiled(plan, v2, xp, n, True, False, act=K.ACT_RELU), ref)              # panel -> row
        assert torch.equal(to_rows(K.spmm_csr_tiled(plan, v2, xp, n, True, True, act=K.ACT_RELU)), ref)      # panel -> panel
        two = K.spmm_csr_tiled(plan, v2, yp, n, True, False, act=K.ACT_NONE)                                # a second hop
        assert torch.equal(two, K.spmm_csr(g.rowptr, g.col, g.val, ref, act=K.ACT_NONE))
    # no plan: a numbering that keeps neighbours close (the panel kernel serves it from L1 / L2 at a lower cost per edge)
    if n >= 40_000:
        rr = np.repeat(np.arange(n), 4)
        cc = np.sort(np.clip(rr.reshape(n, 4) + rng.integers(-50, 51, (n, 4)), 0, n - 1), axis=1).reshape(-1)
        rp = np.arange(0, 4 * n + 1, 4, dtype=np.int64)
        assert CSRGraph(_t(rp, dev), _t(cc.astype(np.int32), dev), _t(np.ones(4 * n, np.float32), dev), n).tile_plan(P) is None
    # no plan: a graph whose columns do not ascend inside a row (the tiled order would not be the CSR order)
    if r.size > 10:
        c2 = c.copy()
        i = int(np.flatnonzero((r[1:] == r[:-1]) & (c[1:] != c[:-1]))[0])
        c2[i], c2[i + 1] = c2[i + 1], c2[i]
        old_near, CSRGraph.TILE_NEAR_ROWS = CSRGraph.TILE_NEAR_ROWS, -1
        try:
            assert CSRGraph(_t(rowptr, dev), _t(c2.astype(np.int32), dev), _t(val, dev), n).tile_plan(P) is None
        finally:
            CSRGraph.TILE_NEAR_ROWS = old_near
