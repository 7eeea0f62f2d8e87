"""The int8 copy's TWO SCALES (csrc/filter_common.h): granules of 32 KiB of int8 rows are NORMAL (quantised on the grid of the
cut) or HEAVY (on the grid of the bank's largest entry), each class with its own measured error and hence its own integer
threshold per query.  The copy's tail and tables against numpy; every reader of the copy -- ring kernel (plain, scored,
pipelined epilogue), direct kernel, single-launch kernel, scored rescoring -- against the fp32 kernels (themselves checked
against the oracle bit for bit) on banks whose heavy rows sit at granule boundaries, whose winners ARE the heavy rows, and
whose granules are all heavy.
"""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _bank(rng, N, D):
    return cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))


def _heavy(kn, rows, rng):
    """Rows with one dominant entry (0.97 of the norm), the rest noise: heavy-tailed, normalised, all different."""
    D = kn.shape[1]
    for r in rows:
        v = 0.03 * rng.standard_normal(D).astype(np.float32)
        v[int(rng.integers(0, D))] = 1.0
        kn[r] = v
    kn[rows] = cref.normalize_rows(kn[rows])
    return kn


def _tables(kb, N, D):
    """(tail words as floats, as ints, granule maxima, class bits) of a bank copy, read back."""
    npad = -(-N // 256) * 256
    gk = 32768 // D
    ngr = -(-npad // gk)
    raw = kb.cpu().numpy().view(np.uint8).reshape(-1)
    t8 = (npad + 1 + npad // 2) * 2 * D
    tail = raw[t8:t8 + 32]
    gm0 = t8 + 2 * D
    gmax = raw[gm0:gm0 + 4 * ngr].view(np.float32)
    c0 = gm0 + (4 * ngr + 15) // 16 * 16
    words = raw[c0:c0 + 4 * ((ngr + 31) // 32)].view(np.uint32)
    bits = np.array([(int(words[g >> 5]) >> (g & 31)) & 1 for g in range(ngr)], dtype=bool)
    return tail.view(np.float32), tail.view(np.int32), gmax, bits, gk, ngr


@pytest.mark.parametrize("D,N", [(256, 20000), (128, 30001), (64, 70000)])
def test_int8_copy_classes_against_numpy(dev, D, N):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(D + N)
    kn = _bank(rng, N, D)
    gk = 32768 // D
    heavy_rows = [5, 3 * gk - 1, 3 * gk, 9 * gk + 17, N - 1]
    kn = _heavy(kn, heavy_rows, rng)
    kb = K.keys_to_bf16(_t(kn, dev))
    f, i, gmax, bits, gk2, ngr = _tables(kb, N, D)
    assert gk2 == gk and int(i[7]) == ngr
    # the granules' maxima, the bank's, the cut and the two scales
    pad = np.zeros((ngr * gk - N, D), np.float32)
    ref_max = np.abs(np.concatenate([kn, pad])).reshape(ngr, -1).max(1)
    assert np.array_equal(gmax, ref_max)
    assert f[2] == ref_max.max() and f[4] == np.float32(f[2]) / np.float32(127.0) and f[1] == np.float32(f[5]) / np.float32(127.0)
    cut = f[5]
    assert np.array_equal(bits, ref_max > cut) and int(i[6]) == int(bits.sum())
    # the heavy rows' granules are heavy ones (a Gaussian bank's own maxima lie far below 0.9), and hardly any other: the
    # model may find that the bank's one or two largest ordinary granules are better off on the coarse grid too
    want = sorted({r // gk for r in heavy_rows})
    got = np.nonzero(bits)[0].tolist()
    assert set(want) <= set(got) and len(got) <= len(want) + max(3, ngr // 16)
    assert ref_max[~bits].max() <= cut < 0.9
    # each class's largest |dk|, from its own grid
    cls_of_row = np.repeat(bits, gk)[:N]
    for heavy, word, sword in ((False, 0, 1), (True, 3, 4)):
        rows = kn[cls_of_row == heavy]
        sk = f[sword]
        ki = np.clip(np.rint(rows * (np.float32(1.0) / sk)), -127, 127).astype(np.float32)
        err2 = ((ki * sk - rows).astype(np.float64) ** 2).sum(1).max()
        assert abs(f[word] - err2) <= 1e-4 * err2 and f[word] >= err2 * (1 - 1e-5)
    c = K.int8_copy_classes(kb, N)
    assert c["heavy_granules"] == len(got) and c["granules"] == ngr and c["cut"] == float(cut)
    assert c["err"] < 0.75 * c["err_heavy"]          # what the second scale is for


def _queries(rng, kn, B, heavy_rows, gk):
    """Random queries, noisy copies of the heavy rows (their winners are heavy keys), noisy copies of keys at granule
    boundaries, one zero query."""
    D = kn.shape[1]
    q = rng.standard_normal((B, D), dtype=np.float32)
    at = 0
    for r in heavy_rows:
        if at + 1 < B:
            q[at] = kn[r] + 0.05 * rng.standard_normal(D).astype(np.float32)
            at += 1
    for g in sorted({r // gk for r in heavy_rows}):
        for r in (g * gk - 1, g * gk, g * gk + gk - 1, g * gk + gk):
            if 0 <= r < kn.shape[0] and at + 1 < B:
                q[at] = kn[r] + 0.3 * rng.standard_normal(D).astype(np.float32)
                at += 1
    if B > 3:
        q[B - 1] = 0.0
    return q


# (B, N, D, k): <= 16 queries take the single-launch kernel through KeyIndex; the others the direct (<= 256) or the ring kernel
_SHAPES = [(2048, 150000, 256, 10),    # ring kernel, pipelined epilogue (D = 256, four groups)
           (6000, 150000, 256, 5),     # ring, scored lists
           (40000, 300000, 128, 10),   # ring, six groups (long streams)
           (5056, 593347, 64, 5),      # ring at D = 64: granules of 512 keys, two int8 levels
           (200, 150000, 256, 10),     # direct kernel, queries in LDS
           (24, 100000, 128, 7),       # direct kernel, queries in registers
           (256, 300000, 64, 5)]       # direct kernel at D = 64


@pytest.mark.parametrize("B,N,D,k", _SHAPES)
@pytest.mark.parametrize("pattern", ["one", "boundaries", "every-granule"])
def test_filtered_topk_on_banks_with_heavy_rows_bit_exact(dev, B, N, D, k, pattern):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(B + N + D + k + len(pattern))
    kn = _bank(rng, N, D)
    gk = 32768 // D
    ngr = -(-N // gk)
    if pattern == "one":
        heavy_rows = [N // 3]
    elif pattern == "boundaries":   # first and last granule, neighbours, alternating runs, both ends of a granule
        gs = [0, 1, 4, 6, 8, 9, 10, ngr // 2, ngr - 2, ngr - 1]
        heavy_rows = sorted({min(N - 1, g * gk + o) for g in gs for o in (0, gk - 1)})
    else:                           # a heavy row in every granule: the cut can only be the maximum -- one class again
        heavy_rows = [min(N - 1, g * gk + int(rng.integers(0, gk))) for g in range(ngr)]
    kn = _heavy(kn, heavy_rows, rng)
    kn[N // 2:N // 2 + 20] = kn[:20]                 # ties
    q = _queries(rng, kn, B, heavy_rows[:12], gk)
    knd, qd = _t(kn, dev), _t(q, dev)
    kb = K.keys_to_bf16(knd)
    c = K.int8_copy_classes(kb, N)
    if pattern == "every-granule":   # (every granule's largest entry is ~0.9: no grid finer than 0.85 / 127 for anybody)
        assert c["cut"] > 0.85 * c["max_abs"] and c["err"] > 0.02
    else:                            # (the model may put a few per cent of the ordinary granules on the coarse grid too)
        assert 0 < c["heavy_granules"] <= len(heavy_rows) + max(4, ngr // 8) and c["err"] < 0.02
    assert K.filtered_i8_levels(B, N, D, k) >= 1
    s1, i1, over = K.topk_cosine_filtered(qd, knd, kb, k, idx_base=3)
    s0, i0 = K.topk_cosine(qd, knd, k, idx_base=3)
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    if pattern != "every-granule":
        assert int(over) <= 1 + B // 100             # (the zero query; nothing the classes would cause)
    # the heavy rows ARE winners of the queries made from them
    first = i1[:min(len(heavy_rows[:12]), B - 1), 0].cpu().numpy() - 3
    assert np.array_equal(first, np.array(heavy_rows[:len(first)]))


@pytest.mark.parametrize("B,N,D,k", [(1, 100000, 256, 10), (4, 200000, 128, 5), (16, 131072, 64, 10), (9, 70000, 256, 32)])
def test_single_launch_kernel_on_banks_with_heavy_rows_bit_exact(dev, B, N, D, k):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(B + N + D + k)
    kn = _bank(rng, N, D)
    gk = 32768 // D
    ngr = -(-N // gk)
    gs = [0, 1, 5, 6, ngr // 2, ngr - 1]
    heavy_rows = sorted({min(N - 1, g * gk + o) for g in gs for o in (0, gk // 2 - 1, gk // 2, gk - 1)})
    kn = _heavy(kn, heavy_rows, rng)
    knd = _t(kn, dev)
    kb = K.keys_to_bf16(knd)
    assert len(gs) <= K.int8_copy_classes(kb, N)["heavy_granules"] <= len(gs) + max(3, ngr // 8)
    for trial in range(3):
        q = _queries(rng, kn, B, heavy_rows[4 * trial:4 * trial + 4], gk) if B > 1 else \
            (kn[heavy_rows[trial]] + 0.05 * rng.standard_normal(D).astype(np.float32))[None]
        qd = _t(q, dev)
        s0, i0 = K.topk_cosine(qd, knd, k, idx_base=7)
        for cap in (-1, 0):          # the library's rule (int8 at D = 128 / 256), then the bf16 copy
            K.set_max_i8_levels(cap)
            try:
                s1, i1, over = K.topk_cosine_small(qd, knd, kb, k, idx_base=7)
            finally:
                K.set_max_i8_levels(-1)
            assert torch.equal(i0, i1) and torch.equal(s0, s1), (trial, cap)


def test_key_index_keeps_a_bank_with_heavy_rows_on_int8(dev):
    """Rounds 3 / 4: one scale for the whole copy, a single one-hot row sent the bank to bf16.  Now the row costs its own
    granule: the index keeps the bank on int8 (the NORMAL granules' measured error decides) and every answer is the
    oracle's; a bank of heavy-tailed rows THROUGHOUT has no class to gain from and stays off int8 as before."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(77)
    N, D, B, k = 70000, 256, 1100, 10
    kn = _bank(rng, N, D)
    kn[123] = 0
    kn[123, 9] = 1.0
    q = rng.standard_normal((B, D), dtype=np.float32)
    q[0] = kn[123] + 0.05 * rng.standard_normal(D).astype(np.float32)
    idx = K.KeyIndex(_t(kn, dev))
    for _ in range(3):
        s, i = idx.topk(_t(q, dev), k)
        torch.cuda.synchronize()
    assert idx._i8_ok is True and not idx._i8_off and not idx._filter_off
    c = idx.i8_classes
    assert 1 <= c["heavy_granules"] <= c["granules"] // 8 and c["err"] <= K.KeyIndex.I8_MAX_ERR < c["err_heavy"] and abs(c["scale_heavy"] - 1 / 127) < 1e-9
    rs, ri = cref.topk_cosine(q[:300], kn, k)
    assert np.array_equal(i.cpu().numpy()[:300], ri) and np.array_equal(s.cpu().numpy()[:300], rs)
    assert int(i[0, 0]) == 123
    # heavy-tailed throughout: every row a dominant entry + noise
    kn2 = _heavy(_bank(rng, N, D), list(range(N)), rng)
    idx2 = K.KeyIndex(_t(kn2, dev))
    s2, i2 = idx2.topk(_t(q, dev), k)
    assert idx2._i8_ok is False
    rs2, ri2 = cref.topk_cosine(q[:200], kn2, k)
    assert np.array_equal(i2.cpu().numpy()[:200], ri2) and np.array_equal(s2.cpu().numpy()[:200], rs2)
    assert K.N.lib().ragraph_topk_cosine_filtered_max_i8_levels(-1) == -1     # the cap does not leak out of a call
