"""World-size-2 gloo test of the sharded retrieval logic (ragraph_amd/sharded.py) on CPU.  The per-shard kernels are
replaced by an oracle-backed `ops` object (tests may use the oracle; the product default is the HIP library), so what
is exercised is the collective path: shard bounds, idx_base, all_gather, canonical merge, owner-sum + all_reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cref


class OracleOps:
    """The five kernels of ragraph_amd.kernels, answered by the CPU checker on torch CPU tensors."""

    @staticmethod
    def normalize_rows(x):
        return torch.from_numpy(cref.normalize_rows(x.numpy()))

    @staticmethod
    def topk_cosine(q, kn, k, idx_base=0):
        s, i = cref.topk_cosine(q.numpy(), kn.numpy(), k, idx_base)
        return torch.from_numpy(s), torch.from_numpy(i)

    @staticmethod
    def topk_merge(s, i):
        a, b = cref.topk_merge(s.numpy(), i.numpy())
        return torch.from_numpy(a), torch.from_numpy(b)

    @staticmethod
    def gather_reduce(v, l, idx, idx_base=0, v_scale=1.0):
        a, b = cref.gather_reduce(v.numpy(), None if l is None else l.numpy(), idx.numpy(), idx_base, v_scale)
        return torch.from_numpy(a), (None if b is None else torch.from_numpy(b))

    @staticmethod
    def gather_rows(v, idx, idx_base=0):
        return torch.from_numpy(cref.gather_rows(v.numpy(), idx.numpy(), idx_base))

    @staticmethod
    def dedup_rows(kn):
        U, largest, uniq, ptr, mem = cref.dedup_rows(kn.numpy())
        return U, largest, torch.from_numpy(uniq), torch.from_numpy(ptr), torch.from_numpy(mem)

    @staticmethod
    def topk_expand_groups(su, iu, ptr, mem, k, idx_base=0, idx_base_u=0):
        s, i = cref.topk_expand_groups(su.numpy(), iu.numpy(), ptr.numpy(), mem.numpy(), k, idx_base, idx_base_u)
        return torch.from_numpy(s), torch.from_numpy(i)

    @staticmethod
    def theta_sharpen(gathered, theta, k):
        G, B, m = gathered.shape
        union = gathered.permute(1, 0, 2).reshape(B, G * m)
        kth = torch.sort(union, dim=1, descending=True).values[:, k - 1]
        torch.maximum(theta, kth, out=theta)
        return theta


class FilteredOracleOps(OracleOps):
    """+ the phased (bound -> level 0 -> level 1) structure of ragraph_topk_cosine_filtered_sharded_f32, restated with
    exact scores: a shard's list holds only what passes the bound the EXCHANGE left in theta.  If an exchange produced an
    invalid (too high) bound, winners would be dropped and the merged result would differ from the single-GPU one."""

    calls = 0
    _prior = None          # ragraph_topk_cosine_filtered_set_prior's thread-local, restated
    spec_calls = 0         # calls that ran under a speculative first bound (no phase 0)

    @classmethod
    def set_filter_prior(cls, p):
        old, cls._prior = cls._prior, p
        return old

    @staticmethod
    def sharded_speculates(B, plan_n, D, k, n_shards):
        return True

    @staticmethod
    def filter_helps(B, n_keys, D, k):
        return True

    @staticmethod
    def keys_to_bf16(kn):
        return kn

    @classmethod
    def topk_cosine_filtered(cls, q, kn, kb, k, idx_base=0, keys_packed=None, exchange=None, plan_n=0):
        cls.calls += 1
        qn = cref.normalize_rows(q.numpy())
        S = cref.cosine_scores(qn, kn.numpy())                       # [B, n_local] exact scores
        B, n = S.shape
        ninf = np.float32(-np.inf)

        def local_topk(lo, hi, theta, prev):
            out_s = np.full((B, k), ninf, np.float32)
            out_i = np.full((B, k), np.iinfo(np.int64).max, np.int64)
            for b in range(B):
                cand = [(S[b, j], j) for j in range(lo, hi) if S[b, j] >= theta[b]]
                if prev is not None:
                    cand += [(prev[0][b, r], int(prev[1][b, r])) for r in range(k) if prev[1][b, r] != np.iinfo(np.int64).max]
                cand.sort(key=lambda t: (-t[0], t[1]))
                for r, (sc, j) in enumerate(cand[:k]):
                    out_s[b, r], out_i[b, r] = sc, j
            return out_s, out_i

        n0 = min(n, max(k, n // 8))
        scores = torch.full((B, k), float("-inf"))
        if cls._prior is not None and exchange is not None:
            # a speculative first bound: theta = the prior for every query, no first sample, NO phase 0 -- a key below the
            # prior is dropped even when it belongs to the true top-k (the owner of the merged row must notice)
            cls.spec_calls += 1
            theta = torch.full((B,), float(cls._prior))
        else:
            theta = torch.from_numpy(np.sort(S[:, :n0], axis=1)[:, ::-1][:, k - 1].copy() if n0 >= k else np.full(B, ninf, np.float32))
            if n0 >= k:  # phase 0: k lower bounds of distinct keys' exact scores (here: the sample's exact top-k)
                scores.copy_(torch.from_numpy(np.sort(S[:, :n0], axis=1)[:, ::-1][:, :k].copy()))
            if exchange is not None:
                exchange(0, theta, scores)
        e1 = n // 2
        s0, i0 = local_topk(0, e1, theta.numpy(), None)
        scores.copy_(torch.from_numpy(s0))
        torch.maximum(theta, scores[:, k - 1], out=theta)
        if exchange is not None:
            exchange(1, theta, scores)
        s1, i1 = local_topk(e1, n, theta.numpy(), (s0, i0))
        i1 = np.where(i1 == np.iinfo(np.int64).max, i1, i1 + idx_base)
        return torch.from_numpy(s1), torch.from_numpy(i1), torch.zeros(1, dtype=torch.int32)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, D, C, B, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ragraph_amd.sharded import ShardedToyGraphBase, shard_bounds

        rng = np.random.default_rng(0)  # same bank on every rank, each keeps its slice
        keys = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
        vals = rng.standard_normal((N, D), dtype=np.float32)
        labs = np.eye(C, dtype=np.float32)[rng.integers(0, C, N)]
        q = rng.standard_normal((B, D), dtype=np.float32)
        lo, hi = shard_bounds(N, world, rank)
        tgb = ShardedToyGraphBase(torch.from_numpy(keys[lo:hi]), torch.from_numpy(vals[lo:hi]),
                                  torch.from_numpy(labs[lo:hi]), lo, k, ops=OracleOps)
        s, i = tgb.topk(torch.from_numpy(q))
        sv, ml, _ = tgb.retrieve_reduced(torch.from_numpy(q))
        e, l = tgb.retrieve(torch.from_numpy(q))
        rep = ShardedToyGraphBase(torch.from_numpy(keys[lo:hi]), torch.from_numpy(vals), torch.from_numpy(labs), lo, k,
                                  ops=OracleOps, values_replicated=True)
        rsv, rml, ri = rep.retrieve_reduced(torch.from_numpy(q))
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), s=s.numpy(), i=i.numpy(), sv=sv.numpy(), ml=ml.numpy(),
                 e=e.numpy(), l=l.numpy(), rsv=rsv.numpy(), rml=rml.numpy(), ri=ri.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,k", [(3001, 10), (12, 10)])  # second case: shards (6 rows) smaller than k
def test_sharded_retrieval_world2_matches_single(tmp_path, N, k):
    D, C, B, world = 64, 3, 37, 2
    mp.spawn(_worker, args=(world, _free_port(), N, D, C, B, k, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    keys = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    vals = rng.standard_normal((N, D), dtype=np.float32)
    labs = np.eye(C, dtype=np.float32)[rng.integers(0, C, N)]
    q = rng.standard_normal((B, D), dtype=np.float32)
    rs, ri = cref.topk_cosine(q, cref.normalize_rows(keys), k)  # the bank object normalises its stored keys once
    rsv, rml = cref.gather_reduce(vals, labs, ri)
    r0, r1 = (dict(np.load(tmp_path / f"r{r}.npz")) for r in range(world))
    for r in (r0, r1):
        assert np.array_equal(r["i"], ri) and np.array_equal(r["s"], rs)  # bit-identical to one GPU, on every rank
        assert np.allclose(r["sv"], rsv, atol=1e-5) and np.array_equal(r["ml"], rml)
        assert np.array_equal(r["e"], vals[ri]) and np.array_equal(r["l"], labs[ri])
        # replicated values: bit-identical to the single-GPU sums, no all_reduce
        assert np.array_equal(r["ri"], ri) and np.array_equal(r["rsv"], rsv) and np.array_equal(r["rml"], rml)
    assert np.array_equal(r0["sv"], r1["sv"])


def _qs_worker(rank, world, port, B, C, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ragraph_amd.sharded import QueryShard

        qs = QueryShard()
        lo, hi = qs.bounds(B)
        full = torch.arange(B * C, dtype=torch.float32).reshape(B, C)  # what one GPU would produce
        got = qs.gather_rows(full[lo:hi].contiguous(), B)               # each rank contributes only its rows
        np.savez(os.path.join(out_dir, f"q{rank}.npz"), got=got.numpy(), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [37, 40, 3])  # ragged split, even split, fewer rows than... no: 3 rows over 2 ranks
def test_query_shard_gather_world2(tmp_path, B):
    """Query sharding: every rank answers its slice of the batch; the all_gather puts the rows back in query order."""
    C, world = 3, 2
    mp.spawn(_qs_worker, args=(world, _free_port(), B, C, str(tmp_path)), nprocs=world, join=True)
    want = np.arange(B * C, dtype=np.float32).reshape(B, C)
    r0, r1 = (dict(np.load(tmp_path / f"q{r}.npz")) for r in range(world))
    assert np.array_equal(r0["got"], want) and np.array_equal(r1["got"], want)
    assert r0["lo"] == 0 and r0["hi"] == r1["lo"] and r1["hi"] == B


def _theta_worker(rank, world, port, N, D, B, k, skew, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ragraph_amd.sharded import ShardedToyGraphBase, shard_bounds

        rng = np.random.default_rng(5)
        keys = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
        q = rng.standard_normal((B, D), dtype=np.float32)
        if skew:  # every winner of the first queries lives in shard 1: shard 0's lists come back empty
            keys[N - 3 * k:] = cref.normalize_rows(q[0:1] + 0.05 * rng.standard_normal((3 * k, D), dtype=np.float32))
        lo, hi = shard_bounds(N, world, rank)
        vals = torch.from_numpy(rng.standard_normal((N, D), dtype=np.float32))
        labs = torch.from_numpy(np.eye(3, dtype=np.float32)[rng.integers(0, 3, N)])
        tgb = ShardedToyGraphBase(torch.from_numpy(keys[lo:hi]), vals, labs, lo, k, ops=FilteredOracleOps,
                                  values_replicated=True)
        assert tgb.plan_n == max(shard_bounds(N, world, r)[1] - shard_bounds(N, world, r)[0] for r in range(world))
        s, i = tgb.topk(torch.from_numpy(q))
        assert FilteredOracleOps.calls == 1
        # the query-sharded tail: this rank's rows only (lists by all_to_all, merge, gathers), then the output all_gather
        tlo, thi = tgb.tail_bounds(B)
        assert (tlo, thi) == shard_bounds(B, world, rank)
        sv, ml, ti = tgb.retrieve_reduced_rows(torch.from_numpy(q))
        full_ml = tgb.gather_output_rows(ml, B)
        np.savez(os.path.join(out_dir, f"t{rank}.npz"), s=s.numpy(), i=i.numpy(), sv=sv.numpy(), ml=ml.numpy(), ti=ti.numpy(),
                 full_ml=full_ml.numpy(), tlo=tlo, thi=thi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("skew", [False, True])
def test_sharded_theta_exchange_world2(tmp_path, skew):
    """Key sharding with bounds sharpened across the shards (all_reduce MAX of the first bound, all_gather of each
    shard's best scores per level): the merged result is the single-GPU one bit for bit, also when one shard holds every
    winner and the other's lists are empty."""
    N, D, B, k, world = 1501, 64, 23, 10, 2
    mp.spawn(_theta_worker, args=(world, _free_port(), N, D, B, k, skew, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    keys = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    q = rng.standard_normal((B, D), dtype=np.float32)
    if skew:
        keys[N - 3 * k:] = cref.normalize_rows(q[0:1] + 0.05 * rng.standard_normal((3 * k, D), dtype=np.float32))
    rs, ri = cref.topk_cosine(q, cref.normalize_rows(keys), k)
    vals = rng.standard_normal((N, D), dtype=np.float32)
    labs = np.eye(3, dtype=np.float32)[rng.integers(0, 3, N)]
    rsv, rml = cref.gather_reduce(vals, labs, ri)
    for r in range(world):
        got = dict(np.load(tmp_path / f"t{r}.npz"))
        assert np.array_equal(got["i"], ri) and np.array_equal(got["s"], rs)
        # query-sharded tail (RAGraph._forward_key_shard): the rank's rows of the single-GPU result, bit for bit, and the
        # gathered [B, C] output in query order on every rank
        lo, hi = int(got["tlo"]), int(got["thi"])
        assert np.array_equal(got["ti"], ri[lo:hi]) and np.array_equal(got["sv"], rsv[lo:hi])
        assert np.array_equal(got["ml"], rml[lo:hi]) and np.array_equal(got["full_ml"], rml)


def _dup_worker(rank, world, port, N, D, B, k, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ragraph_amd.sharded import ShardedToyGraphBase, shard_bounds

        keys, q = _dup_bank(N, D, B)
        lo, hi = shard_bounds(N, world, rank)
        rng = np.random.default_rng(11)
        vals = torch.from_numpy(rng.standard_normal((N, D), dtype=np.float32))
        labs = torch.from_numpy(np.eye(3, dtype=np.float32)[rng.integers(0, 3, N)])
        tgb = ShardedToyGraphBase(torch.from_numpy(keys[lo:hi]), vals, labs, lo, k, ops=FilteredOracleOps, values_replicated=True)
        s, i = tgb.topk(torch.from_numpy(q))
        sv, ml, ti = tgb.retrieve_reduced_rows(torch.from_numpy(q))
        stats = tgb._index.duplicate_stats
        np.savez(os.path.join(out_dir, f"d{rank}.npz"), s=s.numpy(), i=i.numpy(), ti=ti.numpy(), plan_n=tgb.plan_n,
                 searched=tgb._index.search_rows(), rows=hi - lo, stats=np.array(stats if stats else (0, 0, 0)))
    finally:
        dist.destroy_process_group()


def _dup_bank(N, D, B):
    """A bank in the reference's proportions (three quarters one vector, repeats among the rest), the first shard with far
    more distinct rows than the second."""
    rng = np.random.default_rng(21)
    real = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    const = cref.normalize_rows(rng.standard_normal((1, D), dtype=np.float32))[0]
    keys = real.copy()
    keys[np.arange(N) % 4 != 0] = const               # 75 % copies of one vector
    keys[N // 2:][np.arange(N - N // 2) % 8 != 0] = const   # the second half: 87.5 %
    rep = rng.random(N) < 0.2
    keys[rep] = keys[rng.integers(0, N, N)[rep]]
    q = rng.standard_normal((B, D), dtype=np.float32)
    q[0] = const + 0.05 * rng.standard_normal(D).astype(np.float32)
    q[1] = 0.0
    return keys, q


def test_sharded_duplicate_bank_world2(tmp_path):
    """Key sharding over a bank of duplicates: every shard collapses ITS exact duplicates (different numbers of unique rows
    per shard), the ranks agree on plan_n = the largest searched row count, the per-shard lists of unique rows are expanded
    to bank rows before the merge -- the single-GPU result over all N rows, bit for bit."""
    N, D, B, k, world = 9000, 32, 19, 10, 2
    mp.spawn(_dup_worker, args=(world, _free_port(), N, D, B, k, str(tmp_path)), nprocs=world, join=True)
    keys, q = _dup_bank(N, D, B)
    rs, ri = cref.topk_cosine(q, cref.normalize_rows(keys), k)
    got = [dict(np.load(tmp_path / f"d{r}.npz")) for r in range(world)]
    for r, g in enumerate(got):
        assert np.array_equal(g["i"], ri) and np.array_equal(g["s"], rs)
        assert g["searched"] < g["rows"] // 2 and g["stats"][1] == g["searched"]      # collapsed
        assert g["plan_n"] == max(int(x["searched"]) for x in got)
    assert got[0]["searched"] > 1.5 * got[1]["searched"]                               # unequal shards
    lo1 = N // 2
    from ragraph_amd.sharded import shard_bounds
    for r, g in enumerate(got):
        a, b = shard_bounds(B, world, r)
        assert np.array_equal(g["ti"], ri[a:b])
    assert lo1 > 0


def _hybrid_worker(rank, world, port, S, N, D, B, k, skew, out_dir):
    import datetime

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from ragraph_amd import sharded as SH
        from ragraph_amd.sharded import HybridLayout, ShardedToyGraphBase

        beats = []
        SH.heartbeat = beats.append
        layout = HybridLayout(S)
        assert (layout.q, layout.s) == divmod(rank, S) and layout.Q * layout.S == world
        rng = np.random.default_rng(5)
        keys = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
        q = rng.standard_normal((B, D), dtype=np.float32)
        if skew:  # every winner of the first queries lives in the LAST key shard
            keys[N - 3 * k:] = cref.normalize_rows(q[0:1] + 0.05 * rng.standard_normal((3 * k, D), dtype=np.float32))
        vals = torch.from_numpy(rng.standard_normal((N, D), dtype=np.float32))
        labs = torch.from_numpy(np.eye(3, dtype=np.float32)[rng.integers(0, 3, N)])
        lo, hi = layout.key_rows(N)
        tgb = ShardedToyGraphBase(torch.from_numpy(keys[lo:hi]), vals, labs, lo, k, group=layout.key_group,
                                  ops=FilteredOracleOps, values_replicated=True)
        qs = layout.query_shard()
        assert tgb.world == S and tgb.rank == layout.s and qs.world == layout.Q and qs.rank == layout.q
        qlo, qhi, rlo, rhi = HybridLayout.rows(qs, tgb, B)
        sv, ml, ti = HybridLayout.retrieve_reduced_rows(qs, tgb, torch.from_numpy(q))
        full_ml = HybridLayout.gather_output_rows(qs, tgb, ml, B)
        full_sv = HybridLayout.gather_output_rows(qs, tgb, sv, B)
        SH.heartbeat = None
        np.savez(os.path.join(out_dir, f"h{rank}.npz"), sv=sv.numpy(), ml=ml.numpy(), ti=ti.numpy(), full_ml=full_ml.numpy(),
                 full_sv=full_sv.numpy(), rows=np.array([qlo + rlo, qlo + rhi]), exchanges=np.array(sorted(tgb.exchange_count.items())),
                 beats=np.array(len(beats)), filtered_calls=np.array(FilteredOracleOps.calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,S,B,skew", [(4, 2, 23, False), (4, 2, 64, True), (4, 4, 23, False), (4, 1, 23, False)])
def test_hybrid_layout_world4_matches_single(tmp_path, world, S, B, skew):
    """Q query groups x S key shards (ragraph_amd.sharded.HybridLayout; 2 x 4 at G = 8 in bench.py) under world-size-4
    gloo: every rank's rows are the single-GPU rows bit for bit, the ranks' rows tile the batch exactly once, both
    all_gathers put the whole [B, C] / [B, D] result on every rank in query order, and the exchanges stay inside a key
    group (same count on its ranks).  S = world and S = 1 are the two pure layouts through the same code."""
    N, D, k = 1501, 64, 10
    mp.spawn(_hybrid_worker, args=(world, _free_port(), S, N, D, B, k, skew, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    keys = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    q = rng.standard_normal((B, D), dtype=np.float32)
    if skew:
        keys[N - 3 * k:] = cref.normalize_rows(q[0:1] + 0.05 * rng.standard_normal((3 * k, D), dtype=np.float32))
    rs, ri = cref.topk_cosine(q, cref.normalize_rows(keys), k)
    vals = rng.standard_normal((N, D), dtype=np.float32)
    labs = np.eye(3, dtype=np.float32)[rng.integers(0, 3, N)]
    rsv, rml = cref.gather_reduce(vals, labs, ri)
    got = [dict(np.load(tmp_path / f"h{r}.npz")) for r in range(world)]
    covered = np.zeros(B, dtype=np.int64)
    for r, g in enumerate(got):
        lo, hi = (int(x) for x in g["rows"])
        covered[lo:hi] += 1
        assert np.array_equal(g["ti"], ri[lo:hi]) and np.array_equal(g["sv"], rsv[lo:hi]) and np.array_equal(g["ml"], rml[lo:hi])
        assert np.array_equal(g["full_ml"], rml) and np.array_equal(g["full_sv"], rsv)
        assert int(g["filtered_calls"]) == 1
        partner = got[(r // S) * S + (r % S + 1) % S]
        assert np.array_equal(g["exchanges"], partner["exchanges"])
        if S > 1:
            assert int(g["beats"]) >= 3   # the exchanges and collectives reported to the heartbeat hook
    assert (covered == 1).all()


def _prior_worker(rank, world, port, S, N, D, B, k, out_dir):
    """Key-sharded (S = world) or hybrid (S < world) retrieval under the group's speculative first bound: the policy's own
    prior after two warm-up calls, then forced priors -- far below every k-th best, just above the lowest k-th best of ONE
    rank's rows (only that rank sees a miss), above everything."""
    import datetime

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from ragraph_amd.sharded import HybridLayout, ShardedToyGraphBase

        layout = HybridLayout(S)
        rng = np.random.default_rng(5)
        keys = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
        vals = torch.from_numpy(rng.standard_normal((N, D), dtype=np.float32))
        labs = torch.from_numpy(np.eye(3, dtype=np.float32)[rng.integers(0, 3, N)])
        lo, hi = layout.key_rows(N)
        tgb = ShardedToyGraphBase(torch.from_numpy(keys[lo:hi]), vals, labs, lo, k, group=layout.key_group,
                                  ops=FilteredOracleOps, values_replicated=True)
        qs = layout.query_shard()
        out = {}

        def run(tag, q):
            qlo, qhi, rlo, rhi = HybridLayout.rows(qs, tgb, B)
            before = (FilteredOracleOps.calls, FilteredOracleOps.spec_calls, tgb.reruns, dict(tgb.exchange_count))
            sv, ml, ti = HybridLayout.retrieve_reduced_rows(qs, tgb, torch.from_numpy(q))
            full_ml = HybridLayout.gather_output_rows(qs, tgb, ml, B)
            ex0 = tgb.exchange_count.get(0, 0) - before[3].get(0, 0)
            out[tag] = np.array([FilteredOracleOps.calls - before[0], FilteredOracleOps.spec_calls - before[1],
                                 tgb.reruns - before[2], ex0])
            out[tag + "_ti"], out[tag + "_ml"], out[tag + "_rows"] = ti.numpy(), full_ml.numpy(), np.array([qlo + rlo, qlo + rhi])

        qs_all = [rng.standard_normal((B, D), dtype=np.float32) for _ in range(4)]
        for c, q in enumerate(qs_all):          # the policy: two calls with a bound pass, then the learnt prior
            run(f"auto{c}", q)
        out["auto_prior"] = np.array([tgb.prior.prior_for(B, k) or np.nan])
        q = qs_all[0]
        rs, _ = cref.topk_cosine(q, keys, k)
        kth = rs[:, k - 1]
        # my key group answers rows [glo, ghi); its ranks own consecutive slices: the forced "one rank misses" prior sits just
        # above the lowest k-th best of the group's LAST rank's rows (and below every other row's, or it is skipped)
        for tag, prior in (("low", float(kth.min()) - 0.05), ("one", float(out_one_prior(kth, qs, tgb, B))), ("high", float(kth.max()) + 0.05)):
            tgb.prior.forced = prior
            run(tag, q)
        tgb.prior.forced = None
        np.savez(os.path.join(out_dir, f"p{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def out_one_prior(kth, qs, tgb, B):
    """A prior that exactly ONE row of this key group's slice misses: midway between its lowest and second-lowest k-th best."""
    qlo, qhi = qs.bounds(B)
    part = np.sort(kth[qlo:qhi])
    return 0.5 * (part[0] + part[1])


@pytest.mark.parametrize("world,S", [(2, 2), (4, 4), (4, 2)])
def test_sharded_speculative_prior_world(tmp_path, world, S):
    """VERDICT round 5, task 1: the key-sharded entry under a speculative first bound.  Whether a call speculates, whether its
    result stands and whether it is repeated are decisions of the whole key group: every rank runs the same number of
    filtered calls, the same exchanges (none of phase 0 under a prior) and the same number of repeats -- also when the prior is
    too high for ONE row that one rank owns --, and every row is the single-GPU row bit for bit."""
    N, D, B, k = 1501, 64, 48, 10     # (24 rows per query group of the 2 x 2 layout: above the policy's smallest batch)
    mp.spawn(_prior_worker, args=(world, _free_port(), S, N, D, B, k, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    keys = cref.normalize_rows(rng.standard_normal((N, D), dtype=np.float32))
    vals = rng.standard_normal((N, D), dtype=np.float32)
    labs = np.eye(3, dtype=np.float32)[rng.integers(0, 3, N)]
    qs_all = [rng.standard_normal((B, D), dtype=np.float32) for _ in range(4)]
    got = [dict(np.load(tmp_path / f"p{r}.npz")) for r in range(world)]
    for tag, q in [(f"auto{c}", qs_all[c]) for c in range(4)] + [(t, qs_all[0]) for t in ("low", "one", "high")]:
        rs, ri = cref.topk_cosine(q, keys, k)
        _, rml = cref.gather_reduce(vals, labs, ri)
        for r, g in enumerate(got):
            lo, hi = (int(x) for x in g[tag + "_rows"])
            assert np.array_equal(g[tag + "_ti"], ri[lo:hi]) and np.array_equal(g[tag + "_ml"], rml), (tag, r)
            partner = got[(r // S) * S + (r % S + 1) % S]
            assert np.array_equal(g[tag], partner[tag]), (tag, r)          # the key group acted as one
    for g in got:
        assert list(g["auto0"]) == [1, 0, 0, 1] and list(g["auto1"]) == [1, 0, 0, 1]      # warm-up: bound pass + phase 0
        assert g["auto2"][1] >= 1 and g["auto2"][3] == g["auto2"][2]                       # speculative: phase 0 only in a repeat
        assert not np.isnan(g["auto_prior"][0])
        assert list(g["low"]) == [1, 1, 0, 0]                                              # stands: one call, no phase 0
        assert list(g["one"]) == [2, 1, 1, 1] and list(g["high"]) == [2, 1, 1, 1]          # missed: repeated once, with a bound pass
