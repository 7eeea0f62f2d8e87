"""Banks of duplicates (the reference's own bank recipe: RAGraph_node/ragraph_utils/ToyGraphBase.py:91-119 draws a
pass's rows with replacement and Augmentation.py:9-20 zeroes the augmented passes' features, so three of four bank rows
are one vector): ragraph_dedup_rows_f32 / ragraph_topk_expand_groups_f32 against the oracle, and KeyIndex's collapsed
search against the search over every row -- bit for bit."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def reference_shaped_bank(rng, N, D, distinct_fraction=0.25, resample=0.3, zero_rows=False):
    """Rows in the reference's proportions: `distinct_fraction` of the rows are real embeddings, a share `resample` of
    them repeats of other real rows (multinomial with replacement), the rest copies of ONE vector (normalize(PReLU(bias)),
    or the zero row when the encoder's bias is zero) -- interleaved per resource graph: 10 real rows, then 30 copies."""
    n_real = int(N * distinct_fraction)
    real = cref.normalize_rows(rng.standard_normal((n_real, D), dtype=np.float32))
    rep = rng.random(n_real) < resample
    src = rng.integers(0, n_real, n_real)
    real[rep] = real[src[rep]]                      # (a repeat of a repeat is a repeat)
    const = np.zeros(D, np.float32) if zero_rows else cref.normalize_rows(rng.standard_normal((1, D), dtype=np.float32))[0]
    kn = np.empty((N, D), np.float32)
    kn[:] = const
    block = np.arange(N) % 40 < 10                  # 10 sampled rows of the original pass, 30 of the augmented passes
    slots = np.flatnonzero(block)[:n_real]
    kn[slots] = real[:slots.size]
    return kn


@pytest.mark.parametrize("N,D", [(1, 8), (300, 64), (5000, 256), (4097, 128), (2500, 20), (70000, 256)])
def test_dedup_rows_matches_oracle(dev, N, D):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(N + D)
    base = cref.normalize_rows(rng.standard_normal((max(N // 3, 1), D), dtype=np.float32))
    kn = base[rng.integers(0, base.shape[0], N)].copy()
    if N > 10:
        kn[5] = 0.0
        kn[9] = -0.0                                # other bits than +0: its own group
        kn[N - 1] = kn[0]
    U, largest, uniq_row, group_ptr, members = K.dedup_rows(_t(kn, dev))
    oU, olargest, ouniq, optr, omem = cref.dedup_rows(kn)
    assert (U, largest) == (oU, olargest)
    assert np.array_equal(uniq_row.cpu().numpy(), ouniq)
    assert np.array_equal(group_ptr.cpu().numpy(), optr)
    assert np.array_equal(members.cpu().numpy(), omem)


@pytest.mark.parametrize("B,k", [(1, 1), (7, 5), (300, 10), (65, 32), (9, 64)])
def test_topk_expand_groups_bit_exact(dev, B, k):
    """Expansion against the oracle and against the search over every row: groups larger and smaller than k, groups whose
    scores tie (a zero query ties everything; a query orthogonal to the coordinates two rows differ in ties those two),
    fewer unique rows than k, and the -inf / INT64_MAX padding of a shard's list."""
    from ragraph_amd import kernels as K

    D = 64
    rng = np.random.default_rng(17 * B + k)
    base = cref.normalize_rows(rng.standard_normal((90, D), dtype=np.float32))
    twin = base[3].copy()
    twin[[0, 1]] = twin[[1, 0]]                     # differs from base[3] in coordinates 0 and 1 only
    base[4] = twin
    pick = np.concatenate([rng.integers(0, 90, 3000), np.repeat([3, 4], 40)])
    rng.shuffle(pick)
    kn = base[pick].copy()
    q = rng.standard_normal((B, D), dtype=np.float32)
    q[0] = 0.0                                      # everything ties at +0
    if B > 2:
        q[1] = base[3]
        q[1, :2] = 0.0                              # blind to coordinates 0 and 1: the copies of rows 3 and 4 tie at the top
        q[2] = 2.0 * base[7]
    U, _, uniq_row, group_ptr, members = K.dedup_rows(_t(kn, dev))
    ku = min(k, U)
    su, iu = cref.topk_cosine(q, kn[uniq_row.cpu().numpy()], ku)
    s, i = K.topk_expand_groups(_t(su, dev), _t(iu, dev), group_ptr, members, k, idx_base=11)
    os_, oi = cref.topk_expand_groups(su, iu, group_ptr.cpu().numpy(), members.cpu().numpy(), k, idx_base=11)
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(s.cpu().numpy(), os_)
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=11)                     # = the search over every row
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    # a shard's list: padded entries are empty groups; fewer than k rows in all -> padded output
    su2, iu2 = su.copy(), iu.copy()
    su2[:, ku // 2:] = -np.inf
    iu2[:, ku // 2:] = np.iinfo(np.int64).max
    s2, i2 = K.topk_expand_groups(_t(su2, dev), _t(iu2, dev), group_ptr, members, k)
    os2, oi2 = cref.topk_expand_groups(su2, iu2, group_ptr.cpu().numpy(), members.cpu().numpy(), k)
    assert np.array_equal(i2.cpu().numpy(), oi2) and np.array_equal(s2.cpu().numpy(), os2)


@pytest.mark.parametrize("N,D,zero_rows", [(80000, 256, False), (80000, 256, True), (300000, 64, False), (131072, 128, False),
                                           (400000, 256, False)])
def test_key_index_collapses_a_reference_shaped_bank(dev, N, D, zero_rows):
    """75 % of the rows one vector + repeats among the rest: KeyIndex searches the unique rows (its inner index never
    overflows, never leaves the filter) and answers with the bits of the search over every row, at every batch size."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(N + D + zero_rows)
    kn = reference_shaped_bank(rng, N, D, zero_rows=zero_rows)
    knd = _t(kn, dev)
    index = K.KeyIndex(knd)
    const = kn[39]
    for B, k in ((1, 3), (40, 10), (300, 10), (5000, 10), (2100, 32)):
        q = rng.standard_normal((B, D), dtype=np.float32)
        q[0] = const + 0.05 * rng.standard_normal(D).astype(np.float32)   # next to the giant group: k copies of it win
        if B > 3:
            q[1] = 0.0
            q[2] = kn[0]
        qd = _t(q, dev)
        s, i = index.topk(qd, k, idx_base=3)
        torch.cuda.synchronize()
        s32, i32 = K.topk_cosine(qd, knd, k, idx_base=3)                  # every row, fp32 kernel over all N rows
        assert torch.equal(i, i32) and torch.equal(s, s32)
        rows = np.arange(min(B, 40))
        rs, ri = cref.topk_cosine(q[rows], kn, k, idx_base=3)
        assert np.array_equal(i.cpu().numpy()[rows], ri) and np.array_equal(s.cpu().numpy()[rows], rs)
    n, U, largest = index.duplicate_stats
    assert n == N and U < 0.3 * N and largest >= 0.7 * N
    inner = index.search_index
    assert inner is not index and inner.keys_normalized.shape[0] == U
    index.topk(qd, 10)
    torch.cuda.synchronize()
    index.topk(qd, 10)                              # (overflow counts arrive one call later)
    assert index.overflowed_queries <= 6 and not inner._filter_off   # (zero queries may be counted)


def test_key_index_leaves_ordinary_banks_alone(dev):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(5)
    kn = cref.normalize_rows(rng.standard_normal((70000, 256), dtype=np.float32))
    kn[100:130] = kn[:30]                           # a few repeats: searched as it is
    index = K.KeyIndex(_t(kn, dev))
    q = rng.standard_normal((33, 256), dtype=np.float32)
    s, i = index.topk(_t(q, dev), 10)
    assert index._collapsed is False and index.duplicate_stats == (70000, 70000 - 30, 2)
    rs, ri = cref.topk_cosine(q, kn, 10)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    # one row stored a few hundred times in an otherwise ordinary bank: collapsed (every query next to it would fill its
    # candidate list with copies)
    kn[2000:2300] = kn[7]
    index2 = K.KeyIndex(_t(kn, dev))
    q[0] = kn[7]
    s2, i2 = index2.topk(_t(q, dev), 10)
    assert index2._collapsed and index2.duplicate_stats[2] == 302   # (rows 7, 107 and the 300 new copies)
    rs2, ri2 = cref.topk_cosine(q, kn, 10)
    assert np.array_equal(i2.cpu().numpy(), ri2) and np.array_equal(s2.cpu().numpy(), rs2)
