"""The speculative first bound of the filtered top-k (ragraph_topk_cosine_filtered_set_prior; VERDICT round 4, task 2): a
call that starts from theta = prior for every query skips its bound pass, PROVES every answer behind its last level and
scans exactly for the queries the prior was too high for -- so the result has the bits of the fp32 kernel for ANY prior.
Forced priors (far too low, just right, in the middle of the queries' k-th best scores, far too high) through the C ABI, and
the product dispatch (KeyIndex) that derives the prior from its calls' statistics and withdraws it after a miss."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _bank(dev, N, D, seed):
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(seed)
    return K.normalize_rows(torch.randn(N, D, device=dev, generator=g)), g


@pytest.mark.parametrize("B,N,D,k", [(64, 200_000, 256, 10), (300, 200_000, 256, 10), (700, 150_000, 128, 5),
                                      (3000, 200_000, 256, 10), (5000, 300_000, 64, 8), (20_000, 120_000, 256, 10)])
def test_forced_priors_are_exact(dev, B, N, D, k):
    from ragraph_amd import kernels as K

    kn, g = _bank(dev, N, D, B + D)
    kb = K.keys_to_bf16(kn)
    q = torch.randn(B, D, device=dev, generator=g)
    s32, i32 = K.topk_cosine(q, kn, k)
    kth = s32[:, k - 1]
    lo, mid, hi = float(kth.min()), float(kth.median()), float(kth.max())
    cases = [("just below every query's k-th best", lo - 0.01, 0),
             ("the median k-th best: about half of the queries miss", mid, int((kth < mid).sum())),
             ("well below: more candidates, no miss", lo - 0.06, 0)]
    if B <= 300:
        cases.append(("above every query's k-th best: every query is scanned", hi + 0.05, B))
    for what, prior, misses in cases:
        old = K.set_filter_prior(prior)
        assert old != old        # (NaN: no prior was set on this thread)
        try:
            s, i, over, st = K.topk_cosine_filtered(q, kn, kb, k, return_stats=True)
        finally:
            K.set_filter_prior(None)
        assert torch.equal(i, i32) and torch.equal(s, s32), what
        w = st.cpu().tolist()
        assert w[0] == K.FILTER_STATS_MAGIC and w[16] == 1 and w[17] == misses, (what, w[16:20])
        assert int(over) >= misses
        if misses < B:
            assert K.ord2f(w[18]) >= min(lo, prior) - 1e-6 and K.ord2f(w[19]) <= hi + 1e-6
    # without a prior: the bound pass, and the same statistics from the call's last launch
    s, i, over, st = K.topk_cosine_filtered(q, kn, kb, k, return_stats=True)
    w = st.cpu().tolist()
    assert torch.equal(i, i32) and torch.equal(s, s32) and int(over) == 0
    assert w[16] == 0 and w[17] == 0 and abs(K.ord2f(w[18]) - lo) < 1e-6 and abs(K.ord2f(w[19]) - hi) < 1e-6


def test_prior_with_zero_queries_duplicates_and_an_oracle_sample(dev):
    """Zero queries (answered without a scan, never judged against the prior), exact duplicate keys at the k-th rank, and a
    sample of the rows against the CPU oracle."""
    from ragraph_amd import kernels as K

    kn, g = _bank(dev, 100_000, 256, 5)
    kn[50_000:50_040] = kn[7]                                   # 41 copies of one key: ties in every list that holds it
    q = torch.randn(400, 256, device=dev, generator=g)
    q[3] = 0.0
    q[100] = kn[7] + 0.01 * torch.randn(256, device=dev, generator=g)
    kb = K.keys_to_bf16(kn)
    s32, i32 = K.topk_cosine(q, kn, 10)
    for prior in (0.2, float(s32[:, 9][s32[:, 9] > 0].min()) - 0.01, 0.28):
        K.set_filter_prior(prior)
        try:
            s, i, over, st = K.topk_cosine_filtered(q, kn, kb, 10, return_stats=True)
        finally:
            K.set_filter_prior(None)
        assert torch.equal(i, i32) and torch.equal(s, s32), prior
    rows = [0, 3, 100, 399]
    rs, ri = cref.topk_cosine(q[rows].cpu().numpy(), kn.cpu().numpy(), 10)
    assert np.array_equal(i[rows].cpu().numpy(), ri) and np.array_equal(s[rows].cpu().numpy(), rs)


def test_key_index_derives_the_prior_and_withdraws_it_after_a_miss(dev):
    from ragraph_amd import kernels as K

    N, D, k = 300_000, 256, 10
    kn, g = _bank(dev, N, D, 21)
    index = K.KeyIndex(kn)
    priors = []
    for c in range(6):
        q = torch.randn(512, D, device=dev, generator=g)
        s, i = index.topk(q, k)
        torch.cuda.synchronize()                                  # (the statistics arrive behind an event: let them)
        priors.append(index.last_prior)
        s32, i32 = K.topk_cosine(q, kn, k)
        assert torch.equal(i, i32) and torch.equal(s, s32)
    assert priors[0] is None and priors[1] is None               # warming up: bound passes
    assert all(p is not None for p in priors[3:])                 # then speculative, from the calls' own statistics
    st = index._spec[k]
    assert st["used"] >= 2 and st["failed"] == 0 and index.overflowed_queries == 0
    kth_low = float(K.topk_cosine(q, kn, k)[0][:, k - 1].min())
    assert priors[-1] < kth_low and priors[-1] > kth_low - 0.08  # below every k-th best seen, by a margin of their spread
    # queries unlike anything seen: half of their mass lies outside the bank's span, their k-th best scores are far lower
    kn2 = kn.clone()
    kn2[:, 128:] = 0.0
    kn2 = K.normalize_rows(kn2)
    index2 = K.KeyIndex(kn2)
    for c in range(4):
        q = torch.randn(512, D, device=dev, generator=g)
        q[:, 128:] = 0.0
        index2.topk(q, k)
        torch.cuda.synchronize()
    assert index2.last_prior is not None
    q = torch.randn(512, D, device=dev, generator=g)              # full-width queries: scores shrink by 1 / sqrt(2)
    s, i = index2.topk(q, k)
    torch.cuda.synchronize()
    s32, i32 = K.topk_cosine(q, kn2, k)
    assert torch.equal(i, i32) and torch.equal(s, s32)            # exact all the same: the misses were scanned
    index2.topk(q[:64].contiguous(), k)                           # (polls the miss count)
    torch.cuda.synchronize()
    st2 = index2._spec[k]
    assert st2["failed"] > 0 and st2["off_at"] is not None        # withdrawn
    assert not index2._filter_off and not index2._i8_off          # ... and the misses were not blamed on the lists
    s, i = index2.topk(q, k)
    assert index2.last_prior is None and torch.equal(i, i32) and torch.equal(s, s32)


def test_a_prior_far_too_high_only_costs_scans(dev):
    """A prior above every score: every query misses and is answered by the sliced exact scan -- the same bits."""
    from ragraph_amd import kernels as K

    kn, g = _bank(dev, 70_000, 256, 3)
    q = torch.randn(8, 256, device=dev, generator=g)
    K.set_filter_prior(0.9)
    try:
        s, i, over = K.topk_cosine_small(q, kn, K.keys_to_bf16(kn), 10)
    finally:
        K.set_filter_prior(None)
    s32, i32 = K.topk_cosine(q, kn, 10)
    assert torch.equal(i, i32) and torch.equal(s, s32) and int(over) == 8


def test_a_loose_prior_costs_one_call_and_nothing_else(dev):
    """A bank whose queries' k-th best scores spread widely (two clusters queried from inside and from between them): the
    prior derived from the lowest of them floods the int8 levels ONCE -- it is withdrawn, and the candidates it let through
    are not held against the int8 copy (round 5: they were; the bank ran on bf16 afterwards, 1.3 - 1.5 x slower)."""
    from ragraph_amd import kernels as K

    N, D, k, B = 200_000, 256, 10, 4096
    g = torch.Generator(device=dev).manual_seed(31)
    c = torch.randn(40, D, device=dev, generator=g)
    kn = K.normalize_rows(c[torch.randint(0, 40, (N,), device=dev, generator=g)] + 1.2 * torch.randn(N, D, device=dev, generator=g))
    index = K.KeyIndex(kn, dedup=False)
    def batch():
        pull = torch.rand(B, 1, device=dev, generator=g) * 2.0
        return c[torch.randint(0, 40, (B,), device=dev, generator=g)] * pull + torch.randn(B, D, device=dev, generator=g)
    for _ in range(8):
        q = batch()
        s, i = index.topk(q, k)
        torch.cuda.synchronize()
    s32, i32 = K.topk_cosine(q, kn, k)
    assert torch.equal(i, i32) and torch.equal(s, s32)
    kth = s32[:, k - 1]
    assert float(kth.max() - kth.min()) > 0.1                       # a wide spread of k-th best scores
    st = index._spec[k]
    assert st["used"] >= 1
    assert not index._i8_off and not index._filter_off              # whatever the prior did, the copies are not blamed
    assert index.overflowed_queries == 0 or st["off_at"] is not None


@pytest.mark.parametrize("B,N,D,k", [(1, 300_000, 256, 10), (7, 200_000, 128, 5), (16, 300_000, 256, 10), (32, 150_000, 64, 8),
                                      (3, 100_000, 256, 32)])
def test_small_kernel_under_forced_priors(dev, B, N, D, k):
    """The single-launch kernel with a speculative first bound (no bound units, no wait; the last workgroup proves every answer
    and lists the misses for the sliced scan launched behind it): exact for any prior, statistics words as the filtered call's."""
    from ragraph_amd import kernels as K

    kn, g = _bank(dev, N, D, 7 * B + D)
    kb = K.keys_to_bf16(kn)
    q = torch.randn(B, D, device=dev, generator=g)
    if B >= 7:
        q[2] = 0.0                                               # a zero query: answered in place, never judged
    s32, i32 = K.topk_cosine(q, kn, k)
    kth = s32[:, k - 1]
    live = kth[kth != 0.0] if B >= 7 else kth
    lo, hi = float(live.min()), float(live.max())
    nz = B - (1 if B >= 7 else 0)
    for what, prior, misses in (("below every k-th best", lo - 0.01, 0), ("far below", lo - 0.1, 0),
                                ("above every k-th best: every query is scanned", hi + 0.02, nz),
                                ("between", 0.5 * (lo + hi), int((live < 0.5 * (lo + hi)).sum()))):
        K.set_filter_prior(prior)
        try:
            s, i, over, st = K.topk_cosine_small(q, kn, kb, k, return_stats=True)
        finally:
            K.set_filter_prior(None)
        assert torch.equal(i, i32) and torch.equal(s, s32), what
        w = st.cpu().tolist()
        assert w[0] == K.FILTER_STATS_MAGIC and w[16] == 1 and w[17] == misses and int(over) == misses, (what, w[14:20], int(over))
    s, i, over, st = K.topk_cosine_small(q, kn, kb, k, return_stats=True)    # no prior: the bound phase, the same words
    w = st.cpu().tolist()
    assert torch.equal(i, i32) and torch.equal(s, s32) and int(over) == 0
    assert w[16] == 0 and w[17] == 0 and abs(K.ord2f(w[18]) - lo) < 1e-6 and abs(K.ord2f(w[19]) - hi) < 1e-6
    assert int(K._small_state_buf(q.device).abs().sum()) == 0


def test_key_index_speculates_on_single_query_calls(dev):
    """Graph classification retrieves ONE query per forward: the index derives its prior from such calls alone."""
    from ragraph_amd import kernels as K

    kn, g = _bank(dev, 400_000, 256, 77)
    index = K.KeyIndex(kn)
    used = 0
    for c in range(80):
        q = torch.randn(1 + c % 3, 256, device=dev, generator=g)
        s, i = index.topk(q, 10)
        torch.cuda.synchronize()
        used += index.last_prior is not None
        s32, i32 = K.topk_cosine(q, kn, 10)
        assert torch.equal(i, i32) and torch.equal(s, s32)
    assert used >= 30 and index.overflowed_queries <= 3          # (a miss now and then is a scan, not an error)


def test_a_capture_after_warm_up_keeps_its_bound_pass(dev):
    """an index that is already warm (a prior in force for eager calls) must not bake that prior into a HIP
    graph -- the captured launches would carry it by value into every replay, and the statistics that could withdraw it are not
    read under capture.  Captured calls keep their bound pass: replays with queries whose k-th best lies far BELOW the warm
    prior are exact AND scan nothing (statistics word 16 = 0: not speculative; word 20 = 0: no query took the exact scan)."""
    from ragraph_amd import kernels as K
    from ragraph_amd.capture import CapturedForward

    N, D, k, B = 200_000, 256, 10, 300
    kn, g = _bank(dev, N, D, 4242)
    index = K.KeyIndex(kn, dedup=False)
    near = lambda: kn[torch.randint(0, N, (B,), device=dev, generator=g)] + 0.02 * torch.randn(B, D, device=dev, generator=g)
    for _ in range(6):                                   # warm: queries next to stored keys, k-th best scores high
        index.topk(near(), k)
        torch.cuda.synchronize()
    index.topk(near(), k)
    assert index.last_prior is not None                   # eager calls do speculate by now
    stats = torch.zeros(32, dtype=torch.int32, device=dev)

    def fwd(q):
        s, i = index.topk(q, k)
        stats.copy_(index.last_stats[:32])               # (the call's statistics words: a view of its workspace)
        return i

    cap = CapturedForward(fwd, near())
    far = torch.randn(B, D, device=dev, generator=g)     # random directions: k-th best around 0.25, far below the prior
    i_rep = cap(far).clone()
    torch.cuda.synchronize()
    w = stats.cpu().tolist()
    s32, i32 = K.topk_cosine(far, kn, k)
    assert torch.equal(i_rep, i32)
    assert w[0] == K.FILTER_STATS_MAGIC and w[16] == 0 and w[20] == 0, w[14:21]
