"""Any embedding width, any tensor layout: the reference's calculate_cosine_similarity / retrieve take every emb_size
(RAGraph_node/ragraph_utils/SimilarityFunctions.py:6-16, ToyGraphBase.py:56-67) and whatever strides torch hands them.
Widths other than 64 / 128 / 256 reach the exact top-k two ways -- the C ABI's score-slab route (any D) and KeyIndex's
zero-padded bank on the fused / filtered kernels (D < 256) -- both BIT-EXACT against the oracle."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _bank(rng, N, D):
    return cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))


@pytest.mark.parametrize("B,N,D,k", [
    (1, 50, 1, 3),           # one column: every score is +-1 or 0, ties break towards the lower row
    (9, 333, 3, 5),
    (33, 2000, 30, 10),      # D % 4 != 0: unaligned rows
    (64, 5000, 32, 4),       # hid_units = 32 (missing #2)
    (300, 9000, 100, 10),
    (17, 4000, 200, 7),
    (40, 3000, 300, 10),     # wider than every fused kernel
    (130, 6000, 512, 10),
    (5, 1500, 1433, 3),      # Cora's raw feature width
    (20, 700, 257, 40),      # 32 < k <= 64 on an odd width
])
def test_topk_cosine_any_width_c_abi(dev, B, N, D, k):
    """ragraph_topk_cosine_f32 itself (no padding: the slab route)."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(B * 7 + D)
    kn = _bank(rng, N, D)
    q = rng.standard_normal((B, D), dtype=np.float32)
    s, i = K.topk_cosine(torch.from_numpy(q).to(dev), torch.from_numpy(kn).to(dev), k, idx_base=11)
    rs, ri = cref.topk_cosine(q, kn, k)
    assert np.array_equal(i.cpu().numpy(), ri + 11)
    assert np.array_equal(s.cpu().numpy(), rs)


@pytest.mark.parametrize("B,N,D,k", [
    (1, 70000, 100, 10),     # padded to 128: the single-launch kernel for a handful of queries
    (16, 70000, 32, 5),      # padded to 64
    (600, 70000, 100, 10),   # the filtered path on the padded bank
    (2100, 40000, 200, 10),  # padded to 256, int8 levels
    (300, 3000, 30, 6),      # small bank: fp32 kernels / slabs on the padded rows
    (700, 5000, 300, 10),    # no padding possible: slabs
    (1, 2000, 512, 3),
])
def test_key_index_any_width(dev, B, N, D, k):
    """The product dispatch: KeyIndex pads a bank narrower than 256 once and stays on the fused / filtered kernels."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(B + N + D)
    kn = _bank(rng, N, D)
    q = rng.standard_normal((B, D), dtype=np.float32)
    index = K.KeyIndex(torch.from_numpy(kn).to(dev))
    assert index._width == (None if D > 256 else K.padded_dim(D))
    rs, ri = cref.topk_cosine(q, kn, k)
    for _ in range(2):       # (the second call runs with the first one's overflow statistics polled)
        s, i = index.topk(torch.from_numpy(q).to(dev), k)
        assert np.array_equal(i.cpu().numpy(), ri)
        assert np.array_equal(s.cpu().numpy(), rs)
    assert index.overflowed_queries == 0


def test_key_index_padded_bank_of_duplicates(dev):
    """Duplicate collapsing sees the padded rows: still the canonical top-k of every row."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(5)
    base = _bank(rng, 900, 100)
    kn = base[rng.integers(0, 900, size=6000)]
    q = rng.standard_normal((40, 100), dtype=np.float32)
    index = K.KeyIndex(torch.from_numpy(kn).to(dev))
    s, i = index.topk(torch.from_numpy(q).to(dev), 10)
    rs, ri = cref.topk_cosine(q, kn, 10)
    assert index._collapsed
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)


def test_non_contiguous_and_non_f32_inputs(dev):
    """Strided views and other float dtypes are accepted as torch accepts them (made contiguous fp32 on the way in)."""
    from ragraph_amd import kernels as K
    from ragraph_amd.ragraph_utils import SimilarityFunctions

    rng = np.random.default_rng(3)
    kn = _bank(rng, 4000, 128)
    q = rng.standard_normal((50, 128), dtype=np.float32)
    knt = torch.from_numpy(np.ascontiguousarray(kn.T)).to(dev).t()          # column-major storage
    wide = torch.from_numpy(np.concatenate([q, q], 1)).to(dev)
    qv = wide[:, :128]                                                       # a slice of a wider tensor
    assert not knt.is_contiguous() and not qv.is_contiguous()
    rs, ri = cref.topk_cosine(q, kn, 10)
    s, i = K.topk_cosine(qv, knt, 10)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    s, i = K.KeyIndex(knt).topk(qv, 10)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    s64, i64 = K.topk_cosine(qv.double(), knt, 10)                           # float64 queries: rounded to fp32 first
    assert np.array_equal(i64.cpu().numpy(), ri)
    sim = SimilarityFunctions.calculate_cosine_similarity(qv, knt)
    # (the reference normalises BOTH operands on every call, SimilarityFunctions.py:8,11: stored unit rows once more)
    assert np.array_equal(sim.cpu().numpy(), cref.cosine_scores(cref.normalize_rows(q), cref.normalize_rows(kn)))
    one = SimilarityFunctions.calculate_cosine_similarity(qv[0], knt)        # 1-D query (graph flavour)
    assert one.shape == (4000,) and np.array_equal(one.cpu().numpy(), sim[0].cpu().numpy())


@pytest.mark.parametrize("hid", [32, 100, 512])
def test_toy_graph_base_retrieve_any_hidden_size(dev, hid):
    """ToyGraphBase.retrieve with hid_units the fused kernels are not written for (round 4: RagraphNativeError)."""
    from ragraph_amd.ragraph_utils import ToyGraphBase

    rng = np.random.default_rng(hid)
    N, C, B = 3000, 4, 37
    keys = _bank(rng, N, hid)
    vals = rng.standard_normal((N, hid), dtype=np.float32)
    labs = np.eye(C, dtype=np.float32)[rng.integers(0, C, N)]
    tgb = ToyGraphBase(None, C, hid, 3, device=dev)
    tgb.add_resources(*(torch.from_numpy(a).to(dev) for a in (keys, vals, labs)))
    q = rng.standard_normal((B, hid), dtype=np.float32)
    e, l = tgb.retrieve(torch.from_numpy(q).to(dev), None, False)
    _, ri = cref.topk_cosine(q, cref.normalize_rows(keys), tgb.retrieve_num)
    assert e.shape == (B, tgb.retrieve_num, hid) and l.shape == (B, tgb.retrieve_num, C)
    assert np.array_equal(e.cpu().numpy(), vals[ri]) and np.array_equal(l.cpu().numpy(), labs[ri])
    sv, ml, idx = tgb.retrieve_reduced(torch.from_numpy(q).to(dev))
    assert np.array_equal(idx.cpu().numpy(), ri)
