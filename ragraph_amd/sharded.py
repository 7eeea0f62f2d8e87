"""Key-bank sharding across the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The reference is single-GPU (no collective call exists in it).

Partition: rank r owns bank rows [base_r, base_r + N_r).  Per retrieve:
  1. every rank runs the fused cosine+top-k kernel on ITS shard with idx_base = base_r      -> [B,k] scores + global ids
  2. ONE all_gather of the (B,k) score and id tensors (B*k*12 bytes per rank: latency-bound, single hop on the
     fully connected xGMI mesh)                                                               -> [G,B,k]
  3. every rank merges the G lists with the canonical order (score desc, id asc)            -> identical on all ranks,
     and identical to the 1-GPU result bit for bit (scores are shard-independent fmaf chains)
  4. values / labels: either REPLICATED on every rank (default in bench.py: 1 GB at 1M x 256 -- nothing next to 288 GB
     of HBM -- so sum_k V[idx], mean_k L[idx] are local gathers, bit-identical to 1 GPU, and the all_gather of step 2 is
     the only collective), or row-sharded like the keys: every rank sums the winners IT owns and one all_reduce(sum)
     of [B, D+C] finishes it (no [B,k,D] traffic; sums then differ from 1 GPU by fp32 re-association, <= 1e-6 rel).
Indices and scores are exact for any G.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


# ---- collectives ------------------------------------------------------------------------------------------------------
# RCCL ("nccl") moves device tensors directly.  Under "gloo" (CPU tests; two ranks sharing ONE GPU in
# tests/test_gpu_two_rank.py -- RCCL refuses two ranks on one device) device tensors are staged through the host: the
# collective itself then runs on CPU copies, and the device work enqueued before it has completed (the copy to the host
# synchronises the stream), which is exactly the ordering the exchange callback of the filtered call relies on.
def _staged(group, *tensors) -> bool:
    return dist.get_backend(group) == "gloo" and any(t.is_cuda for t in tensors)


class CollectiveTimes:
    """Optional per-collective timing (bench.py, N > 1): while enabled, every collective of this module is bracketed by
    events on the current stream (the stream the collective is ordered on); ms() sums them per kind after a synchronise."""

    def __init__(self):
        self.enabled = False
        self.events = {}   # kind -> [(start, end, bytes)]

    def start(self, kind: str, tensor: torch.Tensor):
        if not (self.enabled and tensor.is_cuda):
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.events.setdefault(kind, []).append((e0, e1, tensor.numel() * tensor.element_size()))
        return e1

    def report(self) -> dict:
        torch.cuda.synchronize()
        return {kind: {"calls": len(ev), "ms": round(sum(a.elapsed_time(b) for a, b, _ in ev), 4),
                       "MB": round(sum(n for _, _, n in ev) / 1e6, 3)} for kind, ev in self.events.items()}

    def reset(self):
        self.events = {}


collective_times = CollectiveTimes()

# Optional progress hook (bench.py, N > 1): a callable taking one short string, called BEFORE every collective of this module
# and at every exchange of a filtered retrieval -- a rank that hangs in a collective (a peer that diverged, a dead link) has
# then named the phase it is waiting in (bench.py writes it to a per-rank heartbeat file that its watchdog prints).
heartbeat = None


def _beat(what: str) -> None:
    if heartbeat is not None:
        heartbeat(what)


def all_gather_into(out: torch.Tensor, inp: torch.Tensor, group=None) -> None:
    _beat(f"all_gather {tuple(out.shape)}")
    end = collective_times.start("all_gather", out)
    if _staged(group, out, inp):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, inp.cpu().contiguous(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)
    if end is not None:
        end.record()


def all_to_all_single(out: torch.Tensor, inp: torch.Tensor, recv, send, group=None) -> None:
    _beat(f"all_to_all {tuple(out.shape)}")
    end = collective_times.start("all_to_all", out)
    if _staged(group, out, inp):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu().contiguous(), recv, send, group=group)
        out.copy_(host)
    else:
        dist.all_to_all_single(out, inp, recv, send, group=group)
    if end is not None:
        end.record()


def all_reduce(t: torch.Tensor, op, group=None) -> None:
    _beat(f"all_reduce {tuple(t.shape)}")
    end = collective_times.start("all_reduce", t)
    if _staged(group, t):
        host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)
    if end is not None:
        end.record()


def shard_bounds(N: int, world: int, rank: int):
    """Contiguous, balanced row ranges: the first N % world ranks hold one extra row."""
    q, r = divmod(N, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class QueryShard:
    """The other way to use G GPUs: the bank (1 GB at 1M x 256 -- nothing next to 288 GB of HBM) is REPLICATED and the
    batch of queries is split, rank r answering rows shard_bounds(B, G, r).  Queries are independent, so there is no
    data-path collective at all; the only exchange is the all_gather of the per-query results (C logits per node),
    which every rank needs only if it wants the whole output.  Results are those of one GPU bit for bit (a score does
    not depend on the query batch).  Key sharding (ShardedToyGraphBase) is for banks that should not be replicated."""

    def __init__(self, group=None, force_collectives: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.collective = self.world > 1 or (force_collectives and dist.is_initialized())

    def bounds(self, B: int):
        return shard_bounds(B, self.world, self.rank)

    def gather_rows(self, local: torch.Tensor, B: int) -> torch.Tensor:
        """[hi-lo, C] per rank -> [B, C] on every rank (rows in query order).  Ranks hold ceil or floor(B / G) rows:
        padded to the larger size for one all_gather_into_tensor, then the padding rows are dropped."""
        if not self.collective:
            return local
        per = -(-B // self.world)
        lo, hi = self.bounds(B)
        buf = local
        if hi - lo < per:
            buf = torch.cat([local, local.new_zeros((per - (hi - lo),) + tuple(local.shape[1:]))], 0)
        out = local.new_empty((self.world * per,) + tuple(local.shape[1:]))
        all_gather_into(out, buf.contiguous(), self.group)
        if self.world * per == B:
            return out
        parts = []
        for r in range(self.world):
            a, b = shard_bounds(B, self.world, r)
            parts.append(out[r * per: r * per + (b - a)])
        return torch.cat(parts, 0)


class HybridLayout:
    """G = Q x S ranks: rank r = q * S + s holds KEY shard s of S and answers QUERY group q of Q -- the layout between the
    two pure ones (S = G: key-sharded, every rank scores all queries against 1 / G of the keys and takes part in G-wide
    exchanges at every phase; S = 1: query-sharded, bank replicated, no exchange).  With S = 2 at G = 8 a rank scores a
    quarter of the queries against half of the keys and has ONE exchange partner: [B / Q, m] bound exchanges and one
    all_to_all of its group's lists with that partner, then the tail on B / G rows; two all_gathers of [., C] outputs (key
    group, then query axis) complete the step.  Every result row is the single-GPU row bit for bit: a score depends neither
    on the query batch nor on the shard (include/ragraph_hip.h, numerics contract).

    Two families of process groups are created here -- by EVERY rank, in the same order (torch.distributed.new_group is a
    collective over the default group): `key_group` = the S ranks of my query group (ShardedToyGraphBase's group),
    `query_group` = the Q ranks that hold my key shard (QueryShard's group for the final gather)."""

    def __init__(self, key_shards: int, timeout=None):
        if not dist.is_initialized():
            raise RuntimeError("HybridLayout: torch.distributed is not initialised")
        world, rank = dist.get_world_size(), dist.get_rank()
        if key_shards < 1 or world % key_shards:
            raise ValueError(f"HybridLayout: {key_shards} key shards do not divide {world} ranks")
        self.S, self.Q = int(key_shards), world // int(key_shards)
        self.q, self.s = divmod(rank, self.S)
        kw = {} if timeout is None else {"timeout": timeout}
        key_groups = [dist.new_group([qq * self.S + ss for ss in range(self.S)], **kw) for qq in range(self.Q)]
        query_groups = [dist.new_group([qq * self.S + ss for qq in range(self.Q)], **kw) for ss in range(self.S)]
        self.key_group, self.query_group = key_groups[self.q], query_groups[self.s]
        self.name = f"hybrid {self.Q}x{self.S} (query groups x key shards)"

    def key_rows(self, N: int):
        """Bank rows of my key shard."""
        return shard_bounds(N, self.S, self.s)

    def query_shard(self) -> "QueryShard":
        return QueryShard(self.query_group, force_collectives=True)

    @staticmethod
    def rows(qs: "QueryShard", tgb: "ShardedToyGraphBase", B: int):
        """(qlo, qhi, lo, hi): my query group answers rows [qlo, qhi) of the batch, and within the group I finish rows
        [qlo + lo, qlo + hi)."""
        qlo, qhi = qs.bounds(B)
        lo, hi = tgb.tail_bounds(qhi - qlo)
        return qlo, qhi, lo, hi

    @staticmethod
    def retrieve_reduced_rows(qs: "QueryShard", tgb: "ShardedToyGraphBase", search_keys, k=None, defer_verify: bool = False):
        """(sum_k V[idx], mean_k L[idx], idx) of MY rows of the batch: my group's slice of the queries against the group's
        key shards (exchanges and the all_to_all stay inside the key group)."""
        qlo, qhi = qs.bounds(search_keys.shape[0])
        return tgb.retrieve_reduced_rows(search_keys[qlo:qhi].contiguous(), k, defer_verify=defer_verify)

    @staticmethod
    def gather_output_rows(qs: "QueryShard", tgb: "ShardedToyGraphBase", local: torch.Tensor, B: int) -> torch.Tensor:
        """[hi - lo, C] per rank -> [B, C] on every rank: the key group's rows first, then the query groups' slices."""
        qlo, qhi = qs.bounds(B)
        return qs.gather_rows(tgb.gather_output_rows(local, qhi - qlo), B)


class GroupPrior:
    """The speculative first bound of a row-sharded bank's filtered calls: ONE decision per call for the whole group of ranks.
    Every rank keeps this object and feeds it the same numbers -- the pooled statistics of each call (all_reduce MAX of
    [misses, -lowest, highest merged k-th best, candidates per query, overflowed lists]) --, so every rank derives the same
    prior, withdraws it at the same call and re-probes at the same call: the number of exchanges of a call (the prior removes
    phase 0) can never differ between ranks.  The policy is KeyIndex's (ragraph_amd/kernels_index.py), on a clock of calls:
    warm after WARM_CALLS calls with a bound pass, prior = lowest seen - max(MARGIN x spread, MIN_MARGIN); a call that reports a
    miss, overflowed lists or a flood of candidates withdraws it for `after` calls (a failed re-probe quadruples that)."""
    WARM_CALLS = 2
    MARGIN = 0.5
    MIN_MARGIN = 0.01
    HISTORY = 16
    REPROBE_CALLS = 64
    MAX_CANDIDATES = 1.5
    MIN_BATCH = 17

    def __init__(self, enabled: bool = True):
        import os
        self.enabled = enabled and os.environ.get("RAGRAPH_SPEC", "1") != "0"
        self.forced = None            # tests: a prior to use whatever the history says (None: the policy)
        self._st = {}
        self.calls = 0                # the clock: filtered calls of the group so far
        self.used = self.missed_calls = 0

    def _state(self, k):
        st = self._st.get(k)
        if st is None:
            st = self._st[k] = {"hist": [], "off_at": None, "after": self.REPROBE_CALLS, "cand": None, "probed_at": None}
        return st

    def prior_for(self, B: int, k: int):
        if self.forced is not None:
            return float(self.forced)
        if not self.enabled or B < self.MIN_BATCH:
            return None
        st = self._state(k)
        if st["off_at"] is not None:
            if self.calls - st["off_at"] < st["after"]:
                return None
            st["off_at"] = None
            st["probed_at"] = self.calls
        if len(st["hist"]) < self.WARM_CALLS:
            return None
        lo = min(h[0] for h in st["hist"])
        hi = max(h[1] for h in st["hist"])
        return lo - max(self.MARGIN * (hi - lo), self.MIN_MARGIN)

    def record(self, k: int, speculative: bool, misses: int, lo: float, hi: float, cand: float, lists_over: int) -> bool:
        """One call's pooled statistics.  Returns True when the call's result stands (no query missed the prior)."""
        self.calls += 1
        st = self._state(k)
        have = lo <= hi and lo != float("inf")
        if lists_over:
            st["hist"] = []
        elif have and not (speculative and misses):
            st["hist"].append((lo, hi))
            del st["hist"][:-self.HISTORY]
        if not speculative:
            if cand >= 0 and not lists_over:
                st["cand"] = cand
            return True
        self.used += 1
        loose = cand >= 0 and st["cand"] is not None and cand > self.MAX_CANDIDATES * max(st["cand"], 32.0)
        if misses or loose or lists_over:
            if st["probed_at"] is not None and self.calls - st["probed_at"] < st["after"]:
                st["after"] = min(st["after"] * 4, 1 << 40)
            st["off_at"] = self.calls
            if misses:
                st["hist"] = []
                self.missed_calls += 1
        return misses == 0


class ShardedToyGraphBase:
    """Retrieval over a row-sharded bank.  `ops` supplies the five kernels (default: ragraph_amd.kernels, i.e. the HIP
    library); the CPU tests inject an oracle-backed object to exercise the collective logic under gloo."""

    def __init__(self, keys, values, labels, idx_base: int, retrieve_num: int, group=None, ops=None,
                 force_collectives: bool = False, values_replicated: bool = False, emulate_world: int = 0,
                 plan_n: int = 0):
        """keys: this rank's key rows.  values/labels: this rank's rows (values_replicated=False) or the WHOLE bank's
        (values_replicated=True).  plan_n: the largest shard's row count (default: all_reduce MAX of the shard sizes).
        emulate_world = G (single process, TIMING ONLY): behave as rank 0 of a G-rank job whose other shards look like
        this one -- the exchanges use this shard's numbers G times; results are not the global top-k."""
        if ops is None:
            from . import kernels as ops  # the HIP library; raises loudly without a GPU
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.idx_base = int(idx_base)
        self.values_replicated = values_replicated
        # run the collectives even in a 1-rank group (exercises the RCCL path on a single-GPU box)
        self.collective = self.world > 1 or (force_collectives and dist.is_initialized())
        self.retrieve_num = retrieve_num
        self.resource_keys, self.resource_values, self.resource_labels = keys, values, labels
        self.keys_normalized = ops.normalize_rows(keys)
        from .kernels_index import KeyIndex  # torch-only helper: copies for the faster kernels, made on first use
        self._index = KeyIndex(self.keys_normalized, ops)
        self.emulate_world = int(emulate_world)
        self._out_shard = None
        self.exchange_count = {}   # phase -> how many exchanges of that phase this rank has taken part in (diagnostic)
        self.prior = GroupPrior()  # the speculative first bound: one decision per call for the whole group
        self._pending = None       # (pinned words, event, k, speculative) of a call whose verification was deferred
        self.reruns = 0            # calls repeated without the prior because a query missed it (diagnostic)
        self._suppress = False     # the next call runs without a prior (it repeats a call that missed)
        self._verify_scores = None

        def exchange_fn(phase, theta, scores):
            return self._exchange(phase, theta, scores)

        exchange_fn.n_shards = self.emulate_world if self.emulate_world > 1 else self.world
        self._exchange_fn = exchange_fn
        # rows this shard SEARCHES: its unique rows when KeyIndex collapses exact duplicates (the reference's bank recipe stores
        # most rows many times over; every shard judges its own rows) -- plan_n is the largest of those over the ranks
        n_local = self._index.search_rows(min_unique=64) if hasattr(self._index, "search_rows") else int(keys.shape[0])
        if plan_n:
            self.plan_n = max(int(plan_n), n_local) if n_local != int(keys.shape[0]) else int(plan_n)
        elif self.collective:
            t = torch.tensor([n_local], dtype=torch.int64, device=keys.device)
            all_reduce(t, dist.ReduceOp.MAX, group)
            self.plan_n = int(t.item())
        else:
            self.plan_n = n_local

    # ---- the exchanges of a filtered retrieval over the sharded bank (kernels.topk_cosine_filtered) ---------------------
    def _exchange(self, phase: int, theta, scores):
        """theta [B]: this shard's lower bound of every query's final k-th best score -> a bound over ALL shards.
        Every phase alike: `scores` [B, k] holds (descending) lower bounds of the exact scores of k distinct keys of this
        shard -- after the bound pass the parts' best approximate scores minus eps, after a level the running exact
        top-k -- and the k-th largest of the union of every shard's best m = 2 ceil(k / G) of them is a lower bound of
        the global k-th best (k-th of a subset): one all_gather of [B, m] scores, 4 m B per query and rank, + a k-th
        selection over [B, G m] (theta_sharpen).  Because phase 0 pools the shards' samples, each shard scans only 1 / G
        of the prefix (n_shards below)."""
        G = self.emulate_world if self.emulate_world > 1 else self.world
        if G <= 1 and not self.collective:
            return
        self.exchange_count[phase] = self.exchange_count.get(phase, 0) + 1
        _beat(f"exchange phase {phase} (#{self.exchange_count[phase]})")
        k = scores.shape[1]
        m = min(k, 2 * (-(-k // G)))
        while G * m > 64 and m > -(-k // G):
            m -= 1
        local = scores[:, :m].contiguous()
        B = local.shape[0]
        if self.emulate_world > 1:
            gathered = local.unsqueeze(0).expand(G, B, m).contiguous()
        else:
            gathered = torch.empty((G, B, m), dtype=local.dtype, device=local.device)
            all_gather_into(gathered.view(G * B, m), local, self.group)
        self.ops.theta_sharpen(gathered, theta, k)

    # ---- the speculative first bound of a sharded call: chosen, proven and withdrawn by the whole group ------------------
    def _choose_prior(self, B: int, k: int):
        """The prior of THIS call (None: a bound pass), the same on every rank: GroupPrior's state is fed pooled numbers
        only, and whether the shape speculates at all comes from the shared plan."""
        G = self.emulate_world if self.emulate_world > 1 else self.world
        spec = getattr(self._index.search_index if hasattr(self._index, "search_index") else self._index, "sharded_speculates", None)
        if self._suppress:          # (the repeat of a call whose prior a query missed: decided by confirm() on every rank alike)
            self._suppress = False
            return None
        if spec is None or not spec(B, k, self.plan_n, G):
            return None
        return self.prior.prior_for(B, k)

    def _post_verify(self, merged_s, k: int, prior, defer: bool) -> bool:
        """Behind the merge: this rank's rows' merged k-th best scores against the prior (a row is proven iff it reaches the
        prior; an all-zero query -- every score +0 -- is answered by index order and needs no proof), the smallest / largest of
        them, this shard's candidates per query and overflowed lists; ONE all_reduce MAX of five floats makes them the group's.
        The words travel to a pinned buffer behind an event: confirm() reads them (at once unless `defer`)."""
        speculative = prior is not None
        if self._verify_scores is not None:   # (emulation: see _topk_rows_once)
            merged_s, self._verify_scores = self._verify_scores, None
        ix = self._index.search_index if hasattr(self._index, "search_index") else self._index
        native = getattr(self.ops, "verify_merged_prior", None)
        if native is not None and merged_s.is_cuda:   # one launch (csrc/topk_filter.hip: verify_merged_prior_kernel)
            st, ov = getattr(ix, "last_stats", None), getattr(ix, "last_over", None)
            words = native(merged_s, prior, st if torch.is_tensor(st) else None, ov if torch.is_tensor(ov) else None)
        else:                                          # the same five numbers with torch ops (the CPU tests' oracle shim)
            kth, top = merged_s[:, k - 1], merged_s[:, 0]
            zero = (top == 0) & (kth == 0)
            ok = zero | (kth >= prior) if speculative else torch.ones_like(zero)
            live = ok & ~zero & (kth > float("-inf"))
            inf = torch.full_like(kth, float("inf"))
            none = torch.tensor(float("-inf"), device=merged_s.device)   # (a rank without rows of this batch)
            words = torch.stack([(~ok).sum().to(torch.float32),
                                 -torch.where(live, kth, inf).min() if kth.numel() else none,
                                 torch.where(live, kth, -inf).max() if kth.numel() else none,
                                 self._cand_per_query(merged_s.device), self._lists_over(merged_s.device)])
        if self.collective:
            all_reduce(words, dist.ReduceOp.MAX, self.group)
        if words.is_cuda:
            if getattr(self, "_host_words", None) is None:
                self._host_words = torch.zeros(5, dtype=torch.float32).pin_memory()
                self._event = torch.cuda.Event()
            self._host_words.copy_(words, non_blocking=True)
            self._event.record()
            self._pending = (self._host_words, self._event, k, speculative)
        else:
            self._pending = (words, None, k, speculative)
        return True if defer else self.confirm()

    def _cand_per_query(self, device):
        ix = self._index.search_index if hasattr(self._index, "search_index") else self._index
        st = getattr(ix, "last_stats", None)
        if st is None or not torch.is_tensor(st) or st.numel() < 8:
            return torch.tensor(-1.0, device=device)
        w = st[:8].to(torch.float32)
        return (w[2:5] / w[5:8].clamp(min=1.0)).sum()

    def _lists_over(self, device):
        ix = self._index.search_index if hasattr(self._index, "search_index") else self._index
        ov = getattr(ix, "last_over", None)
        if ov is None or not torch.is_tensor(ov):
            return torch.tensor(0.0, device=device)
        return ov.reshape(-1)[0].to(device=device, dtype=torch.float32)

    def confirm(self) -> bool:
        """The pooled verdict on the last sharded call: True = its result stands.  False = a query's merged k-th best fell
        below the speculative prior somewhere in the group: EVERY rank sees the same words, withdraws the prior and repeats
        the call (the caller does: topk / topk_rows themselves unless asked to defer)."""
        p = self._pending
        if p is None:
            return True
        self._pending = None
        if p[1] is not None:
            p[1].synchronize()
        w = [float(x) for x in p[0].tolist()]
        if self.emulate_world > 1:
            w[0] = 0.0   # (timing only: the same launches, no verdict -- an emulated rank's lists are one shard's, not the group's)
        ok = self.prior.record(p[2], p[3], int(w[0]), -w[1], w[2], w[3], int(w[4]))
        if not ok:
            self.reruns += 1
            self._suppress = True
        return ok

    def topk(self, search_keys, k=None):
        """Global canonical top-k: (scores [B,k], idx [B,k]) identical on every rank."""
        k = self.retrieve_num if k is None else k
        q = search_keys.reshape(1, -1) if search_keys.dim() == 1 else search_keys
        n_local = self.keys_normalized.shape[0]
        kl = min(k, n_local)
        sharded = self.collective or self.emulate_world > 1
        exch = self._exchange_fn if sharded and kl == k else None
        prior = self._choose_prior(q.shape[0], k) if exch is not None else None
        s, i = self._topk_all(q, k, kl, exch, prior)
        if exch is not None and not self._post_verify(s, k, prior, defer=False):
            self._suppress = False
            s, i = self._topk_all(q, k, kl, exch, None)   # (withdrawn on every rank alike: a bound pass)
            self._post_verify(s, k, None, defer=False)
        return s, i

    def _topk_all(self, q, k, kl, exch, prior):
        sharded = self.collective or self.emulate_world > 1
        s, i = self._index.topk(q, kl, idx_base=self.idx_base, exchange=exch, plan_n=self.plan_n, prior=prior)
        if kl < k:  # a shard smaller than k: pad with sentinels that lose every comparison
            pad_s = torch.full((q.shape[0], k - kl), float("-inf"), dtype=s.dtype, device=s.device)
            pad_i = torch.full((q.shape[0], k - kl), torch.iinfo(torch.int64).max, dtype=i.dtype, device=i.device)
            s, i = torch.cat([s, pad_s], 1), torch.cat([i, pad_i], 1)
        if self.emulate_world > 1:  # (timing only) the merge launch a real rank would run over the G gathered lists
            G = self.emulate_world
            self._verify_scores, _ = self.ops.topk_merge(s.unsqueeze(0).expand(G, *s.shape).contiguous(),
                                                         i.unsqueeze(0).expand(G, *i.shape).contiguous())
            return s, i
        if not self.collective:
            return s, i
        # shards that cannot hold k winners of a query pad with sentinels that lose every comparison (-inf, INT64_MAX)
        B = s.shape[0]
        gs = torch.empty((self.world * B, k), dtype=s.dtype, device=s.device)   # rank-major concatenation
        gi = torch.empty((self.world * B, k), dtype=i.dtype, device=i.device)
        all_gather_into(gs, s.contiguous(), self.group)
        all_gather_into(gi, i.contiguous(), self.group)
        return self.ops.topk_merge(gs.view(self.world, B, k), gi.view(self.world, B, k))

    # ---- key-sharded bank, query-sharded tail -----------------------------------------------------------------------
    # Everything behind the per-shard filter is per query: merge, value / label gather, fusion, decoder.  With the
    # values replicated, rank r finishes only rows shard_bounds(B, G, r) of the batch: the per-shard lists travel by ONE
    # all_to_all (a rank receives its rows' lists from every shard: B k 12 (G - 1) / G bytes instead of the all_gather's
    # B k 12 (G - 1)), the merge, the gathers and the decoder run on B / G rows, and one all_gather of the [B, C] outputs
    # (RAGraph.forward) completes the step.  Same bits as the replicated tail: every step is per row.
    def tail_bounds(self, B: int):
        G = self.emulate_world if self.emulate_world > 1 else self.world
        return shard_bounds(B, G, 0 if self.emulate_world > 1 else self.rank)

    def topk_rows(self, search_keys, k=None, defer_verify: bool = False):
        """Global canonical top-k of THIS RANK'S rows of the batch: (scores, idx) [hi - lo, k], (lo, hi) = tail_bounds.
        defer_verify: the caller enqueues more work first and calls confirm() itself before it hands out anything -- when that
        returns False it repeats this call (RAGraph._forward_key_shard / _forward_hybrid)."""
        k = self.retrieve_num if k is None else k
        q = search_keys.reshape(1, -1) if search_keys.dim() == 1 else search_keys
        B = q.shape[0]
        lo, hi = self.tail_bounds(B)
        n_local = self.keys_normalized.shape[0]
        sharded = self.collective or self.emulate_world > 1
        if not sharded or min(k, n_local) < k:
            s, i = self.topk(q, k)
            return s[lo:hi].contiguous(), i[lo:hi].contiguous()
        prior = self._choose_prior(B, k)
        ms, mi = self._topk_rows_once(q, k, lo, hi, prior)
        if not self._post_verify(ms, k, prior, defer=defer_verify):
            self._suppress = False
            ms, mi = self._topk_rows_once(q, k, lo, hi, None)   # (withdrawn on every rank alike: a bound pass)
            self._post_verify(ms, k, None, defer=False)
        return ms, mi

    def _topk_rows_once(self, q, k, lo, hi, prior):
        B = q.shape[0]
        s, i = self._index.topk(q, k, idx_base=self.idx_base, exchange=self._exchange_fn, plan_n=self.plan_n, prior=prior)
        if self.emulate_world > 1:  # (timing only) the merge a real rank runs over the G lists of its rows
            G = self.emulate_world
            ss, ii = s[lo:hi], i[lo:hi]
            # (the verdict on the prior is taken on what the merge returns, as on a real rank: G copies of this shard's list
            # stand in for the other shards' -- a shard's own list under the pooled bound is far shorter than k)
            self._verify_scores, _ = self.ops.topk_merge(ss.unsqueeze(0).expand(G, *ss.shape).contiguous(),
                                                         ii.unsqueeze(0).expand(G, *ii.shape).contiguous())
            return ss.contiguous(), ii.contiguous()
        G = self.world
        bounds = [shard_bounds(B, G, r) for r in range(G)]
        send = [b - a for a, b in bounds]                    # rows of my lists that go to rank r (its slice of the batch)
        recv = [hi - lo] * G                                 # every shard sends me its lists for my rows
        gs = torch.empty((G * (hi - lo), k), dtype=s.dtype, device=s.device)   # shard-major
        gi = torch.empty((G * (hi - lo), k), dtype=i.dtype, device=i.device)
        all_to_all_single(gs, s.contiguous(), recv, send, self.group)
        all_to_all_single(gi, i.contiguous(), recv, send, self.group)
        return self.ops.topk_merge(gs.view(G, hi - lo, k), gi.view(G, hi - lo, k))

    def retrieve_reduced_rows(self, search_keys, k=None, defer_verify: bool = False):
        """(sum_k V[idx], mean_k L[idx], idx) for this rank's rows of the batch (values replicated)."""
        if not self.values_replicated:
            raise ValueError("retrieve_reduced_rows: needs the values / labels replicated on every rank")
        _, idx = self.topk_rows(search_keys, k, defer_verify=defer_verify)
        sum_v, mean_l = self.ops.gather_reduce(self.resource_values, self.resource_labels, idx)
        return sum_v, mean_l, idx

    def gather_output_rows(self, local: torch.Tensor, B: int) -> torch.Tensor:
        """[hi - lo, C] per rank -> [B, C] on every rank (QueryShard.gather_rows over this bank's group)."""
        if self.emulate_world > 1 or not self.collective:
            return local
        if self._out_shard is None:
            self._out_shard = QueryShard(self.group, force_collectives=True)
        return self._out_shard.gather_rows(local, B)

    def retrieve_reduced(self, search_keys, k=None):
        """(sum_k V[idx], mean_k L[idx], idx) over the whole bank."""
        k = self.retrieve_num if k is None else k
        _, idx = self.topk(search_keys, k)
        if self.values_replicated:  # every winner's row is local: same kernel, same bits as a single GPU
            sum_v, mean_l = self.ops.gather_reduce(self.resource_values, self.resource_labels, idx)
            return sum_v, mean_l, idx
        sum_v, _ = self.ops.gather_reduce(self.resource_values, None, idx, idx_base=self.idx_base)
        sum_l, _ = self.ops.gather_reduce(self.resource_labels, None, idx, idx_base=self.idx_base)
        if self.collective:
            packed = torch.cat([sum_v, sum_l], dim=1)
            all_reduce(packed, dist.ReduceOp.SUM, self.group)
            sum_v, sum_l = packed[:, :sum_v.shape[1]].contiguous(), packed[:, sum_v.shape[1]:].contiguous()
        return sum_v, sum_l / float(k), idx  # label counts are small integers: exact sums, one rounded division

    def retrieve(self, search_keys, search_adj=None, add_noise=False):
        """Public (B,k,D)/(B,k,C) form of ToyGraphBase.retrieve: owner-gather + all_reduce (zeros elsewhere)."""
        _, idx = self.topk(search_keys, self.retrieve_num)
        if self.values_replicated:
            return self.ops.gather_rows(self.resource_values, idx), self.ops.gather_rows(self.resource_labels, idx)
        e = self.ops.gather_rows(self.resource_values, idx, idx_base=self.idx_base)
        l = self.ops.gather_rows(self.resource_labels, idx, idx_base=self.idx_base)
        if self.collective:
            all_reduce(e, dist.ReduceOp.SUM, self.group)
            all_reduce(l, dist.ReduceOp.SUM, self.group)
        return e, l
