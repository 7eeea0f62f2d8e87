"""Parity at BASELINE.json's full sizes (1M x 256 bank, k=10; 4M x 64 edge bank), where the CPU oracle can only
afford a sample: (a) the oracle on a random subset of the queries, bit-exact; (b) size-independent properties of the
whole result -- query-batch independence (a score is one fmaf chain, so B, the tile/split plan and the kernel variant
must not change any bit), shard-merge invariance (2 / 3 / 8 row shards merged == the single-shard result),
descending canonical order, indices unique and in range, idempotence."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bank1m(dev):
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(1234)
    kn = K.normalize_rows(torch.randn(1_000_000, 256, device=dev, generator=g))
    q = torch.randn(4096, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
    return kn, q


def test_full_bank_oracle_sample_and_properties(dev, bank1m):
    from ragraph_amd import kernels as K

    kn, q = bank1m
    k = 10
    s, i = K.topk_cosine(q, kn, k)
    # (a) oracle on 48 sampled queries against the whole 1M-key bank
    sel = torch.randperm(q.shape[0], generator=torch.Generator().manual_seed(0))[:48]
    rs, ri = cref.topk_cosine(q[sel.to(dev)].cpu().numpy(), kn.cpu().numpy(), k)
    assert np.array_equal(i[sel.to(dev)].cpu().numpy(), ri)
    assert np.array_equal(s[sel.to(dev)].cpu().numpy(), rs)
    # (b) properties of the full result
    assert bool((s[:, :-1] >= s[:, 1:]).all())
    assert int(i.min()) >= 0 and int(i.max()) < kn.shape[0]
    srt = torch.sort(i, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    s2, i2 = K.topk_cosine(q, kn, k)
    assert torch.equal(s, s2) and torch.equal(i, i2)  # idempotent / deterministic (no atomics anywhere)
    # query-batch independence across kernel variants and split plans: B = 1..128 (streaming kernel, 1-8 groups of 16
    # queries, with and without the pre-pass), 129, 300, 1000 (tile kernel, 1-4 query tiles)
    for lo, B in [(5, 1), (100, 7), (32, 16), (64, 17), (500, 100), (700, 128), (800, 129), (1000, 300), (2000, 1000)]:
        sb, ib = K.topk_cosine(q[lo:lo + B].contiguous(), kn, k)
        assert torch.equal(ib, i[lo:lo + B]) and torch.equal(sb, s[lo:lo + B]), f"B={B} differs from the 4096 batch"
    # smaller k is a prefix of larger k
    s5, i5 = K.topk_cosine(q[:512].contiguous(), kn, 5)
    assert torch.equal(i5, i[:512, :5]) and torch.equal(s5, s[:512, :5])


@pytest.mark.parametrize("G", [2, 3, 8])
def test_full_bank_shard_merge_invariance(dev, bank1m, G):
    from ragraph_amd import kernels as K
    from ragraph_amd.sharded import shard_bounds

    kn, q = bank1m
    qq = q[:1024].contiguous()
    full_s, full_i = K.topk_cosine(qq, kn, 10)
    ss, ii = [], []
    for r in range(G):
        lo, hi = shard_bounds(kn.shape[0], G, r)
        s, i = K.topk_cosine(qq, kn[lo:hi], 10, idx_base=lo)
        ss.append(s)
        ii.append(i)
    ms, mi = K.topk_merge(torch.stack(ss), torch.stack(ii))
    assert torch.equal(mi, full_i) and torch.equal(ms, full_s)


def test_edge_bank_4m_x64(dev):
    """Config 5 shape: 4M x 64 bank, slabs of 4096 queries (RAGraph_edge/modules/RAGraph.py:45,298)."""
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(7)
    kn = K.normalize_rows(torch.randn(4_000_000, 64, device=dev, generator=g))
    q = torch.randn(4096, 64, device=dev, generator=g)
    s, i = K.topk_cosine(q, kn, 10)
    sel = slice(100, 132)
    rs, ri = cref.topk_cosine(q[sel].cpu().numpy(), kn.cpu().numpy(), 10)
    assert np.array_equal(i[sel].cpu().numpy(), ri) and np.array_equal(s[sel].cpu().numpy(), rs)
    s1, i1 = K.topk_cosine(q[7:8].contiguous(), kn, 10)  # small-batch kernel, D=64
    assert torch.equal(i1, i[7:8]) and torch.equal(s1, s[7:8])
    v = torch.randn(4_000_000, 64, device=dev, generator=g)
    mean_v, _ = K.gather_reduce(v, None, i, v_scale=0.1)
    ref, _ = cref.gather_reduce(v.cpu().numpy(), None, i[:64].cpu().numpy(), v_scale=0.1)
    assert np.array_equal(mean_v[:64].cpu().numpy(), ref)


def test_gnn_100k_nodes_matches_oracle(dev, monkeypatch):
    """Config-2 graph (100k nodes, ~1.1M non-zeros, F = 128 -> D = 256): GCN layer + 3-hop propagation.  The oracle keeps the
    REFERENCE's association A_hat (X W^T); the HIP inference path re-associates this shape to (A_hat X) W^T:
      * RAGRAPH_GCN_REFERENCE_ORDER=1: bit-exact against the oracle in the reference order;
      * default: bit-exact against the oracle's re-associated form (the kernels compute what they claim) AND within 1e-5 of
        the reference order, with a tie-aware top-k check: where the two embeddings give different top-k rows for a query,
        the scores at the disagreeing ranks are within 1e-5 of each other (a near-tie the rounding may flip)."""
    from oracle import cref, pipeline
    from ragraph_amd import kernels as K
    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.ragraph_utils import Propagation

    n, F = 100_000, 128
    torch.manual_seed(0)
    adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, seed=8, device=dev), n)
    X = torch.randn(n, F, device=dev)
    pre = PrePrompt(F, 256, "prelu", 1, 0.3).to(dev)
    conv = pre.gcn.convs[0]
    csr = (adj.rowptr.cpu().numpy(), adj.col.cpu().numpy(), adj.val.cpu().numpy())
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
        args = (X.cpu().numpy(), csr, conv.fc.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy(), float(conv.act.weight))
        monkeypatch.setenv("RAGRAPH_GCN_REFERENCE_ORDER", "1")
        h_ref = pre.inference(X, adj)
        y_ref = Propagation.aggregate_k_hop_features(adj, h_ref, 3)
        monkeypatch.delenv("RAGRAPH_GCN_REFERENCE_ORDER")
        h = pre.inference(X, adj)
        y = Propagation.aggregate_k_hop_features(adj, h, 3)
    oh_ref = pipeline.gcn_layer(*args)                               # the reference's order
    assert np.array_equal(h_ref.cpu().numpy(), oh_ref)
    assert np.array_equal(y_ref.cpu().numpy(), pipeline.propagate(csr, oh_ref, 3))
    oh = pipeline.gcn_layer(*args, order="aggregate_first")
    assert pipeline.aggregate_first_applies(F, 256) and not np.array_equal(oh, oh_ref)
    assert np.array_equal(h.cpu().numpy(), oh)
    assert np.array_equal(y.cpu().numpy(), pipeline.propagate(csr, oh, 3))
    assert float((h - h_ref).abs().max()) <= 1e-5 and float((y - y_ref).abs().max()) <= 1e-5
    # tie-aware top-k: 2000 of the nodes as queries against a 50 000-key bank, both embeddings
    g = torch.Generator(device=dev).manual_seed(3)
    kn = K.normalize_rows(torch.randn(50_000, 256, device=dev, generator=g))
    rows = torch.arange(0, n, 50, device=dev)
    s_a, i_a = K.topk_cosine(h[rows].contiguous(), kn, 10)
    s_r, i_r = K.topk_cosine(h_ref[rows].contiguous(), kn, 10)
    differ = (i_a != i_r)
    assert float(differ.float().mean()) < 0.01
    assert float((s_a - s_r).abs().max()) <= 1e-5                   # rank by rank the scores agree to rounding
    if bool(differ.any()):
        assert float((s_a[differ] - s_r[differ]).abs().max()) <= 1e-5
    # row-stochastic propagation of a constant stays constant (size-independent sanity property)
    ones = torch.ones(n, 256, device=dev)
    assert torch.allclose(Propagation.aggregate_k_hop_features(adj, ones, 2), ones, atol=1e-5)


@pytest.mark.gpu
def test_filtered_topk_matches_fp32_kernel_at_full_size(dev):
    """c2-sized bank (1M x 256): the bf16-filtered exact top-k returns the same bits as the fp32 kernel for 20k queries,
    and as the oracle on a sample of them."""
    from oracle import cref
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(11)
    kn = K.normalize_rows(torch.randn(1_000_000, 256, device=dev, generator=g))
    q = torch.randn(20_000, 256, device=dev, generator=g)
    assert K.filter_helps(q.shape[0], kn.shape[0], 256, 10)
    s1, i1, over = K.topk_cosine_filtered(q, kn, K.keys_to_bf16(kn), 10, idx_base=5, keys_packed=K.pack_keys(kn))
    s0, i0 = K.topk_cosine(q, kn, 10, idx_base=5)
    assert torch.equal(i0, i1) and torch.equal(s0, s1)
    assert over == 0  # an ordinary bank never needs the overflow path
    rows = torch.arange(0, 20_000, 1250)
    rs, ri = cref.topk_cosine(q[rows].cpu().numpy(), kn.cpu().numpy(), 10, idx_base=5)
    assert np.array_equal(i1[rows].cpu().numpy(), ri) and np.array_equal(s1[rows].cpu().numpy(), rs)


@pytest.mark.gpu
def test_reference_recipe_bank_1m_rows(dev):
    """A 1M x 256 bank built by the reference's own recipe (bank_build.build_toy_graph over 25 000 synthetic resource
    graphs; RAGraph_node/ragraph_utils/ToyGraphBase.py:91-119, Augmentation.py:9-20): three quarters of its rows are one
    vector and the sampled rows repeat.  The product dispatch collapses the exact duplicates, searches the unique rows on
    the filtered path WITHOUT overflowing, and returns the bits of the fp32 kernel over all 1M rows (every query) and of
    the oracle (a sample)."""
    from ragraph_amd import kernels as K
    from ragraph_amd.bank_build import build_reference_recipe_bank
    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt

    F, C, D, k, N = 128, 3, 256, 10, 1_000_000
    torch.manual_seed(0)
    pre = PrePrompt(F, D, "prelu", 1, 0.3).to(dev)
    with torch.no_grad():
        pre.gcn.convs[0].bias.normal_(0, 0.1)           # (a pre-trained encoder's bias is not zero)
        tgb = build_reference_recipe_bank(pre, N, F, C, D, device=dev)
        n_q = 20_000
        adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n_q, 10, seed=8, device=dev), n_q)
        h = pre.inference(torch.randn(n_q, F, device=dev, generator=torch.Generator(device=dev).manual_seed(4321)), adj)
    kn = tgb.keys_normalized
    assert kn.shape == (N, D)
    s, i = tgb.topk(h, k)
    index = tgb._index
    n, U, largest = index.duplicate_stats
    assert n == N and U <= N // 2 and largest >= N // 2     # >= 50 % duplicate rows; one vector stored > 500 000 times
    inner = index.search_index
    assert inner is not index and inner.keys_normalized.shape[0] == U and U >= 65536
    s0, i0 = K.topk_cosine(h, kn, k)                        # the fp32 kernel over every one of the 1M rows
    assert torch.equal(i, i0) and torch.equal(s, s0)
    rows = torch.arange(0, n_q, 1250)
    rs, ri = cref.topk_cosine(h[rows].cpu().numpy(), kn.cpu().numpy(), k)
    assert np.array_equal(i[rows].cpu().numpy(), ri) and np.array_equal(s[rows].cpu().numpy(), rs)
    # the search over the unique rows took the filtered path and nothing overflowed
    assert K.filter_helps(n_q, U, D, k) and inner._bf16 is not None and not inner._filter_off
    _, _, over = K.topk_cosine_filtered(h, inner.keys_normalized, inner._bf16, k)
    assert int(over) == 0
    # single queries and a few hundred (the reference's real batch sizes) through the same index: same rows
    for lo, B in ((7, 1), (100, 16), (1000, 500)):
        sb, ib = tgb.topk(h[lo:lo + B].contiguous(), k)
        assert torch.equal(ib, i[lo:lo + B]) and torch.equal(sb, s[lo:lo + B])
    torch.cuda.synchronize()
    tgb.topk(h[:64].contiguous(), k)
    assert index.overflowed_queries == 0


@pytest.mark.gpu
def test_filtered_topk_beyond_2gib_of_bf16_keys(dev):
    """4.5M x 256 keys: the bf16 copy (2.3 GB) and the fp32 / packed copies (4.6 GB each) cross the 2^31- and 2^32-byte
    marks, where a 32-bit or sign-extended offset in a kernel's address arithmetic would read the wrong rows (seen once:
    a readfirstlane'd low word sign-extended into the DMA base).  All three exact paths must agree bit for bit, without
    the overflow fallback, and the winners must really come from the whole bank."""
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(12)
    n = 4_500_000
    kn = K.normalize_rows(torch.randn(n, 256, device=dev, generator=g))
    kb, kp = K.keys_to_bf16(kn), K.pack_keys(kn)
    for B in (16, 300, 700):  # one query group per wave / two; streaming vs tile fp32 kernel; wide rescoring
        q = torch.randn(B, 256, device=dev, generator=g)
        q[0] = kn[n - 1] + 0.01 * q[0]   # its best key is the bank's last row
        q[1] = kn[n // 2 + 77]           # ... and one just past the 2 GiB mark of the bf16 copy
        s1, i1, over = K.topk_cosine_filtered(q, kn, kb, 10, idx_base=3)
        s0, i0 = K.topk_cosine(q, kn, 10, idx_base=3)
        assert over == 0
        assert torch.equal(i0, i1) and torch.equal(s0, s1)
        if K.packed_keys_help(B, 256, 10):
            s2, i2 = K.topk_cosine(q, kn, 10, idx_base=3, keys_packed=kp)
            assert torch.equal(i0, i2) and torch.equal(s0, s2)
        assert int(i1[0, 0]) == n - 1 + 3 and int(i1[1, 0]) == n // 2 + 77 + 3
    del kn, kb, kp
    torch.cuda.empty_cache()
    # the edge flavour's D = 64 with four query groups per wave (B >= 1024): 20M keys = 2.56 GB of bf16
    n = 20_000_000
    kn = K.normalize_rows(torch.randn(n, 64, device=dev, generator=g))
    kb = K.keys_to_bf16(kn)
    q = torch.randn(2048, 64, device=dev, generator=g)
    q[0] = kn[n - 1]
    q[1] = kn[n // 2 + 12345]
    s1, i1, over = K.topk_cosine_filtered(q, kn, kb, 10, idx_base=3)
    s0, i0 = K.topk_cosine(q, kn, 10, idx_base=3)
    assert over == 0 and torch.equal(i0, i1) and torch.equal(s0, s1)
    assert int(i1[0, 0]) == n - 1 + 3 and int(i1[1, 0]) == n // 2 + 12345 + 3


@pytest.mark.gpu
@pytest.mark.parametrize("G,B,pool", [(2, 3000, True), (4, 20000, True), (3, 200, True), (4, 3000, False)])
def test_sharded_filtered_exchange_matches_single_gpu(dev, G, B, pool):
    """ragraph_topk_cosine_filtered_sharded_f32 on G row shards of a 1M x 256 bank, one host thread and one stream per
    shard on this one GPU, the per-phase exchanges done through a thread barrier exactly as ShardedToyGraphBase does
    them with RCCL (k-th of the union of every shard's best m values at every phase; pool = False: the weaker contract
    of the header -- all_reduce MAX of the first bound, n_shards = 1, every shard scanning the whole first sample):
    merged lists == the unsharded call, bit for bit; and the shards' candidate work shrinks (lists padded with -inf)."""
    import threading

    from ragraph_amd import kernels as K
    from ragraph_amd.sharded import shard_bounds

    g = torch.Generator(device=dev).manual_seed(21)
    N, D, k = 1_000_000, 256, 10
    kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    q = torch.randn(B, D, device=dev, generator=g)
    q[0] = kn[N - 1]                       # a winner in the last shard's last row
    q[1] = kn[5] + 0.02 * q[1]             # all of this query's best keys near one row of shard 0
    full_s, full_i = K.topk_cosine(q, kn, k)
    bounds = [shard_bounds(N, G, r) for r in range(G)]
    plan_n = max(hi - lo for lo, hi in bounds)
    shards = [kn[lo:hi].contiguous() for lo, hi in bounds]
    copies = [K.keys_to_bf16(s) for s in shards]
    torch.cuda.synchronize()
    barrier = threading.Barrier(G)
    slots = [None] * G
    out, errs, phases = [None] * G, [], []
    m = min(k, 2 * (-(-k // G)))

    def exchange_for(r):
        def exchange(phase, theta, scores):
            phases.append(phase)
            torch.cuda.current_stream().synchronize()          # this shard's numbers are final
            slots[r] = theta.clone() if (phase == 0 and not pool) else scores[:, :m].clone()
            torch.cuda.current_stream().synchronize()
            barrier.wait()
            if phase == 0 and not pool:
                theta.copy_(torch.stack(slots).max(dim=0).values)
            else:
                K.theta_sharpen(torch.stack(slots).contiguous(), theta, k)   # [G, B, m], as an all_gather leaves it
            torch.cuda.current_stream().synchronize()
            barrier.wait()
        exchange.n_shards = G if pool else 1
        return exchange

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                s, i, over = K.topk_cosine_filtered(q, shards[r], copies[r], k, idx_base=bounds[r][0],
                                                    exchange=exchange_for(r), plan_n=plan_n)
                torch.cuda.current_stream().synchronize()
                out[r] = (s, i, int(over))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    ms, mi = K.topk_merge(torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out]))
    assert torch.equal(mi, full_i) and torch.equal(ms, full_s)
    assert all(o[2] == 0 for o in out)
    assert phases and all(p0 == 0 for p0 in phases[:G])   # every shard went through the first-bound exchange


@pytest.mark.gpu
@pytest.mark.parametrize("B,prior_mode", [(3000, None), (300, None), (3000, "low"), (3000, "mid"), (300, "low")])
def test_sharded_unequal_shards_of_a_duplicate_bank(dev, B, prior_mode):
    """Three row shards of a bank of duplicates, each collapsed by its own KeyIndex to a different number of unique rows: one
    searched as it is (200 000 distinct rows = plan_n), one at a quarter of that (the same phases over proportionally
    fewer keys), one tiny (an exact participant: its fp32 top-k offered at every exchange).  Threads as ranks, exchanges
    through a barrier as in the test above; the expanded per-shard lists merge to the fp32 kernel's result over all
    600 000 rows, bit for bit.  prior_mode (round 6): the same under the group's speculative first bound -- every shard's thread
    sets the same prior, no phase-0 exchange happens (the short shard, an exact participant, skips it too) -- "low": below every
    query's k-th best, every row proven; "mid": at the 30 % quantile, where the owner's verdict must name exactly the rows whose
    true k-th best is below it and every other row must carry the fp32 kernel's bits."""
    import threading

    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(33)
    D, k, G, n = 256, 10, 3, 200_000
    const = K.normalize_rows(torch.randn(1, D, device=dev, generator=g))
    shards = []
    for r, distinct in enumerate((1.0, 0.25, 0.05)):
        rows = K.normalize_rows(torch.randn(n, D, device=dev, generator=g))
        if distinct < 1.0:
            keep = torch.rand(n, device=dev, generator=g) < distinct
            rows[~keep] = const
            rep = torch.rand(n, device=dev, generator=g) < 0.1
            rows[rep] = rows[torch.randint(0, n, (n,), device=dev, generator=g)[rep]]
        shards.append(rows.contiguous())
    kn = torch.cat(shards)
    q = torch.randn(B, D, device=dev, generator=g)
    q[0] = const[0] + 0.05 * q[0]          # next to the row stored ~340 000 times in shards 1 and 2
    q[1] = shards[2][7]
    q[2] = 0.0
    full_s, full_i = K.topk_cosine(q, kn, k)
    idx = [K.KeyIndex(sh) for sh in shards]
    searched = [ix.search_rows(min_unique=64) for ix in idx]
    assert searched[0] == n and 40_000 < searched[1] < 70_000 and searched[2] < 16_000
    plan_n = max(searched)
    prior = None
    if prior_mode is not None:
        assert idx[0].sharded_speculates(B, k, plan_n, G)
        kth = full_s[:, k - 1]
        live = kth[full_s[:, 0] != 0]
        prior = float(live.min()) - 0.01 if prior_mode == "low" else float(torch.quantile(live, 0.3))
    barrier = threading.Barrier(G)
    slots = [None] * G
    out, errs = [None] * G, []
    phases_seen = []
    m = min(k, 2 * (-(-k // G)))

    def exchange_for(r):
        def exchange(phase, theta, scores):
            phases_seen.append(phase)
            torch.cuda.current_stream().synchronize()
            slots[r] = scores[:, :m].clone()
            torch.cuda.current_stream().synchronize()
            barrier.wait()
            K.theta_sharpen(torch.stack(slots).contiguous(), theta, k)
            torch.cuda.current_stream().synchronize()
            barrier.wait()
        exchange.n_shards = G
        return exchange

    def run(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                s, i = idx[r].topk(q, k, idx_base=r * n, exchange=exchange_for(r), plan_n=plan_n, prior=prior)
                torch.cuda.current_stream().synchronize()
                out[r] = (s, i)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    ms, mi = K.topk_merge(torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out]))
    if prior is None:
        assert torch.equal(mi, full_i) and torch.equal(ms, full_s)
        assert 0 in phases_seen
        return
    assert 0 not in phases_seen and len(phases_seen) % G == 0        # no first-bound exchange on ANY shard, the rest line up
    words = K.verify_merged_prior(ms, prior).cpu().tolist()
    zero = (full_s[:, 0] == 0) & (full_s[:, k - 1] == 0)
    proven = zero | (full_s[:, k - 1] >= prior)
    differs = (mi != full_i).any(dim=1) | (ms != full_s).any(dim=1)
    assert not bool((differs & proven).any())                          # every proven row: the fp32 kernel's bits
    assert int(words[0]) == int((~proven).sum())                       # the verdict names exactly the others
    if prior_mode == "low":
        assert int(words[0]) == 0 and not bool(differs.any())
    else:
        assert int(words[0]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("tool,args", [("soak_filtered.py", ["8", "123"]), ("soak_filtered.py", ["8", "124", "index"]),
                                       ("shard_soak.py", ["soak", "8", "125"]), ("soak_ops.py", ["8", "126"])])
def test_randomised_soaks_short(dev, tool, args):
    """A few seconds of each randomised parity soak (tools/): random shapes of the filtered top-k (the C entry and the
    KeyIndex dispatch), of 2-8 emulated key shards, and of the dense / sparse / row kernels against the fp32 kernel / the
    CPU oracle.  The long runs found what the fixed shapes had missed (DESIGN.md section 0, row "soak"); this keeps them
    running.  (A child process: the soaks own their streams and threads.)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", tool)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok:" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("F,order", [(128, None), (128, "1"), (300, None)])
def test_inference_rows_are_the_rows_of_inference(dev, monkeypatch, F, order):
    """PrePrompt.inference_rows (a query-sharded rank encodes ITS rows before its retrieval starts; the whole-graph encode runs
    beside it): rows [lo, hi) of inference(), bit for bit -- the aggregate-first association (F = 128 -> 256), the reference's
    order (RAGRAPH_GCN_REFERENCE_ORDER=1), a width that keeps the reference order anyway (300), and two layers."""
    from ragraph_amd import data
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt

    if order:
        monkeypatch.setenv("RAGRAPH_GCN_REFERENCE_ORDER", order)
    n, D = 30_000, 256
    torch.manual_seed(1)
    g = CSRGraph.from_edge_index_sym_normalized(data.synthetic_big_graph(n, 10, seed=3, device=dev), n)
    x = torch.randn(n, F, device=dev)
    for layers in (1, 2):
        pre = PrePrompt(F, D, "prelu", layers, 0.3).to(dev)
        whole = pre.inference(x, g)
        for lo, hi in ((0, 3750), (11_000, 19_001), (n - 1, n)):
            part = pre.inference_rows(x, g, lo, hi)
            assert part.shape == (hi - lo, D) and torch.equal(part, whole[lo:hi])
