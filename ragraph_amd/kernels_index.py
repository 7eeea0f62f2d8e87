"""Bank-side dispatch of the exact top-k (torch only: importable on a CPU box for the gloo tests)."""
from __future__ import annotations

import os

import torch


class KeyIndex:
    """One bank version on the device: the normalised keys plus the copies the faster kernels stream (packed fp32 for
    the LDS-DMA ring, bf16 for the filter), made on first use, and the dispatch to the fastest exact top-k for a batch.
    `ops` supplies the kernels (default: this module); an object without the optional entries (the CPU tests' oracle
    shim) simply always takes topk_cosine."""

    MAX_FILTERED_BATCH = 262144
    # int8 levels only for banks whose int8 copy is accurate enough: max |dk| of the dequantised rows.  Gaussian-like unit
    # rows of 256 elements give 0.015 (eps ~ 0.022, ~3x the bf16 candidates: pays); the error grows
    # with the bank's largest entry (ONE scale for all keys), and a bank with a one-hot row reaches 0.036 -- thousands of
    # candidates per query.  Above this the levels stay on bf16; a bank that still overflows on int8 is caught by
    # _poll_overflow.
    I8_MAX_ERR = 0.02
    OVERFLOW_FRACTION = 1.0 / 64   # of a call of >= 64 queries sent to the exact scan: the filter (level dtype) costs more than it saves
    # Exact duplicates (the reference's bank recipe stores three of four rows as copies of ONE vector, ToyGraphBase.py:91-119
    # + Augmentation.py:9-20; the sampled rows repeat as well, :98 replacement=True): banks of at least DEDUP_MIN_ROWS rows
    # are grouped once per bank version (hash + sort + bit-wise compare on the device, one synchronisation), and when at
    # most DEDUP_MAX_UNIQUE of the rows are unique -- or one row is stored more than DEDUP_MAX_GROUP times: every query
    # next to it would fill its candidate list with copies -- the search runs over the unique rows (their own copies,
    # their own dispatch) and the winners are expanded to bank rows in canonical order: the same bits.
    # Candidates per query the int8 levels of a call may pass (sampled by the call itself, read back asynchronously): an int8
    # level saves ~0.2 ps of matrix time per (query, key) against bf16 and passes ~3x the candidates at ~0.25 ns each, so
    # 2/3 of its candidates must stay below ~0.8e-3 x its keys; with a floor for short levels.  A bank beyond that is
    # slower on int8 than on bf16 although nothing overflows (near-duplicate clusters of a few hundred rows): bf16 from
    # then on.
    # (measured, 17 000 queries x 70 000 keys: an ordinary bank passes 117 + 102 on its two int8 levels -- 54 + 41 on bf16 --, a
    # bank with a 3000-key cluster around the queries 665 -- 64 on bf16 --, nothing overflowing: tools/stats_probe.py)
    I8_MAX_CANDIDATES_BASE = 300.0
    I8_MAX_CANDIDATES_PER_KEY = 1.2e-3
    # A demotion (int8 -> bf16, filter -> fp32 kernels) is a verdict on the QUERIES seen so far as much as on the bank: after
    # this many more queries the bank is tried one level up again, on a call of at most REPROBE_MAX_BATCH queries (a probe that
    # overflows costs an exact scan per overflowed query); a probe that fails demotes again and quadruples the interval.
    REPROBE_QUERIES = 1 << 20
    REPROBE_MAX_BATCH = 4096
    # Speculative first bound (ops.set_filter_prior, csrc/topk_filter.hip): every filtered call reports the smallest and the
    # largest final k-th best score of its queries; once SPEC_WARM_CALLS calls over SPEC_WARM_QUERIES queries have, the next
    # calls of at least SPEC_MIN_BATCH queries start from prior = lowest seen - max(SPEC_MARGIN x (highest - lowest seen),
    # SPEC_MIN_MARGIN) instead of running a bound pass (a fifth to a quarter of a mid-sized call).  The call proves every
    # answer and scans exactly for the queries the prior was too high for: one such query costs more than the pass saves, so
    # a call that reports misses -- or candidates beyond SPEC_MAX_CANDIDATES x the bound pass's calls' -- withdraws the
    # prior until the re-probe interval has passed (a failed re-probe quadruples it).  Banks whose queries' k-th best
    # scores spread widely (a prior far below most of them: a flood of candidates) end there after one call.
    SPEC_MIN_BATCH = 17
    SPEC_WARM_CALLS = 2
    SPEC_WARM_QUERIES = 64
    SPEC_MARGIN = 0.5
    SPEC_MIN_MARGIN = 0.01
    SPEC_MAX_CANDIDATES = 1.5
    SPEC_HISTORY = 16
    DEDUP_MIN_ROWS = 2048
    DEDUP_MAX_UNIQUE = 0.9
    DEDUP_MAX_GROUP = 64

    def __init__(self, keys_normalized: torch.Tensor, ops=None, dedup: bool = True):
        if ops is None:
            from . import kernels as ops  # the HIP library; raises loudly without a GPU
        self.ops = ops
        # Row widths the fused kernels are not written for (the reference takes any emb_size, SimilarityFunctions.py:6-16):
        # up to 256 the bank is zero-padded ONCE to the next fused width and every query per call -- zero columns change no
        # bit of a norm or a score (kernels.padded_dim) --; wider banks take the score-slab path of topk_cosine (any D).
        self.dim = int(keys_normalized.shape[1])
        pad = getattr(ops, "padded_dim", None)
        self._width = pad(self.dim) if pad is not None else self.dim   # None: no fused kernel for this width
        if self._width is not None and self._width != self.dim:
            keys_normalized = ops.pad_cols(keys_normalized, self._width)
        self.keys_normalized = keys_normalized
        self._packed = None
        self._bf16 = None
        self._filter_off = False  # set when this bank defeats the filter (see _poll_overflow)
        self._i8_off = False      # set when this bank defeats the INT8 levels only (clustered keys: too many within the int8 bound)
        self.i8_classes = None    # the int8 copy's two classes of granules as read for _i8_ok (kernels.int8_copy_classes)
        self._pending = None      # (pinned word, event, batch, had int8 levels) of the last filtered call's overflow count
        self._i8_ok = None        # the int8 copy's error row read (once, lazily): accurate enough for int8 levels?
        self._seen_i8, self._seen_bf16 = [0, 0], [0, 0]   # [queries, overflowed] of the polled calls with / without int8
        self._queries = 0                                  # queries answered so far (the clock of the re-probes)
        self._demoted = {"i8": None, "filter": None}       # query count at the demotion, or None
        self._reprobe_after = {"i8": self.REPROBE_QUERIES, "filter": self.REPROBE_QUERIES}
        self._probed_at = {"i8": None, "filter": None}     # query count at the last re-probe
        self._host_word = self._event = None
        self._overflowed = 0
        self.last_i8_candidates = None   # candidates per query over the int8 levels of the last polled call (sampled)
        # speculative first bound, per k: {"hist": [(lowest, highest k-th best of a polled call)], "queries": seen, "off_at":
        # query count at which it was withdrawn (None: in use), "after": re-probe interval, "cand": candidates per query of
        # the last polled call WITH a bound pass, "failed": misses so far}
        self._spec = {}
        self.spec_enabled = os.environ.get("RAGRAPH_SPEC", "1") != "0"   # (RAGRAPH_SPEC=0: every call with its bound pass -- A/B)
        self._notes = 0                  # calls offered to _note_overflow (small calls report every fourth once settled)
        self.last_stats = None           # device view of the last filtered call's statistics words (this index, this stream)
        self.last_prior = None           # the speculative first bound the last filtered call ran with (None: a bound pass)
        self.last_over = None            # sharded calls: the last call's overflow count (device int; the owner of the bank reads it)
        # None: duplicates not looked at yet; False: looked at, searched as it is; else (KeyIndex over the unique rows,
        # group_ptr, members)
        self._collapsed = None if dedup else False
        self.duplicate_stats = None   # (rows, unique rows, largest group) once judged

    @property
    def overflowed_queries(self) -> int:
        """Queries of polled filtered calls that went to the exact scan (diagnostic)."""
        return self._overflowed + (self._collapsed[0].overflowed_queries if self._collapsed else 0)

    @property
    def search_index(self) -> "KeyIndex":
        """The index whose rows the kernels actually stream: the one over the unique rows when the bank was collapsed."""
        return self._collapsed[0] if self._collapsed else self

    def search_rows(self, min_unique: int = 0) -> int:
        """Rows the kernels search for this bank: judges its duplicates now (ShardedToyGraphBase calls this on every rank
        before the ranks agree on plan_n).  A collapsed bank of fewer than min_unique unique rows is searched as it is."""
        if self._collapsed is None:
            self._judge_duplicates()
        if self._collapsed and self._collapsed[0].keys_normalized.shape[0] < min_unique:
            self._collapsed = False
        return int(self.search_index.keys_normalized.shape[0])

    def _judge_duplicates(self):
        """Once per bank version (never while a HIP graph is being captured: one read-back)."""
        kn = self.keys_normalized
        dedup = getattr(self.ops, "dedup_rows", None)
        if dedup is None or kn.shape[0] < self.DEDUP_MIN_ROWS:
            self._collapsed = False
            return
        if kn.is_cuda and torch.cuda.is_current_stream_capturing():
            return   # (stays undecided: searched as it is for now)
        U, largest, uniq_row, group_ptr, members = dedup(kn)
        self.duplicate_stats = (kn.shape[0], U, largest)
        if U > self.DEDUP_MAX_UNIQUE * kn.shape[0] and largest <= self.DEDUP_MAX_GROUP:
            self._collapsed = False
            return
        unique = self.ops.gather_rows(kn, uniq_row)   # (normalised rows copied bit for bit: still normalised)
        self._collapsed = (KeyIndex(unique, self.ops, dedup=False), group_ptr, members)

    def _poll_overflow(self):
        """The filtered call repairs overflowed rows on the device and reads nothing back; its count arrives here after
        the fact (pinned host word + event, polled without waiting).  A bank of near-duplicates (thousands of keys within
        the bound of a query's k-th best) sends its queries to the exact fallback scan: still exact, but a scan reads the
        whole bank for ONE query (~0.2 ms of HBM time at 1M x 256 keys, 1.7 ms when it is the only one) where the fp32
        kernels spend ~3.5 us more per query than the filter: beyond OVERFLOW_FRACTION of a batch the levels first leave
        int8 (the bf16 bound is ~5x tighter), then the bank leaves the filter."""
        pend = self._pending
        if pend is None or (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            return  # (an event query is not a capturable call)
        if not pend[1].query():
            return
        n_over, B = int(pend[0][0]), pend[2]   # (no kernel counts all-zero queries: they are answered without a scan)
        if n_over < 0:                          # (the statistics words came in one copy: their word 20 is the count)
            n_over = int(pend[0][21]) if int(pend[0][1]) == 0x52414753 else 0
        self._pending = None
        words = pend[0][1:].tolist()
        speculative = False
        if len(pend) > 4 and pend[4] and len(words) >= 20 and words[0] == 0x52414753:
            speculative = bool(words[16])
            n_over = self._judge_prior(pend[4], B, words, n_over)   # (what is left is the lists' fault)
        self._overflowed += n_over
        i8_was_off = self._i8_off               # (the call ran under this setting: the overflow rule below judges IT)
        # the call's sampled candidate counts (int8 levels only are judged -- and only of a call with a bound pass: what a
        # prior far below the queries' k-th best lets through says nothing about the int8 copy, _judge_prior withdraws the prior)
        if pend[3] and pend[0].numel() > 16 and not speculative:
            from .kernels import filter_stats_levels
            i8 = [(keys, c) for dt, keys, c in filter_stats_levels(pend[0][1:17].tolist()) if dt == "int8" and c is not None]
            if i8:
                self.last_i8_candidates = sum(c for _, c in i8)
                if self.last_i8_candidates > self.I8_MAX_CANDIDATES_BASE + self.I8_MAX_CANDIDATES_PER_KEY * sum(kk for kk, _ in i8):
                    self._demote("i8")
        # judged over whole calls of >= 64 queries, or over the calls seen so far once they add up to 8 queries (graph
        # classification retrieves ONE query per forward: a bank that sends every such call to the exact scan must not stay)
        acc = self._seen_i8 if pend[3] else self._seen_bf16
        acc[0] += B
        acc[1] += n_over
        if (B >= 64 and n_over >= 2 and n_over > self.OVERFLOW_FRACTION * B) or (acc[0] >= 8 and 4 * acc[1] > acc[0]):
            if pend[3] and not i8_was_off:     # the call(s) had int8 levels: their wider bound is the first suspect
                self._demote("i8")
            else:
                self._demote("filter")
            acc[0] = acc[1] = 0
        elif acc[0] >= 4096:
            acc[0] = acc[1] = 0

    def _judge_prior(self, k: int, B: int, words, n_over: int) -> int:
        """A polled call's statistics words: feed the k-th-best history; judge a speculative call (misses, candidates,
        overflowed lists).  Returns the overflow count the LISTS are to be judged by: a speculative call's misses are not
        their fault, and neither are lists that overflowed under a prior (far below most queries' k-th best it floods them:
        the prior goes, the bank is judged by its calls with a bound pass)."""
        st = self._spec_state(k)
        spec, failed, lo, hi = words[16], words[17], words[18], words[19]
        lists_over = max(n_over - failed, 0)
        cand = None
        if any(words[5 + l] for l in range(3)):
            cand = sum(words[2 + l] / words[5 + l] for l in range(min(words[1], 3)) if words[5 + l])
        if lists_over:
            st["hist"], st["queries"] = [], 0      # (a bank whose lists overflow is no ground for a prior)
        elif lo != 0x7FFFFFFF and hi != -0x80000000 and B - failed > 0:
            from .kernels import ord2f
            st["hist"].append((ord2f(lo), ord2f(hi)))
            del st["hist"][:-self.SPEC_HISTORY]
            st["queries"] += B
        if not spec:
            if cand is not None and not lists_over:
                st["cand"] = cand
            return lists_over
        st["used"] += 1
        st["failed"] += failed
        loose = cand is not None and st["cand"] is not None and cand > self.SPEC_MAX_CANDIDATES * max(st["cand"], 32.0)
        if failed or loose or lists_over:
            probed_at = st.get("probed_at")
            if probed_at is not None and self._queries - probed_at < st["after"]:
                st["after"] = min(st["after"] * 4, 1 << 40)   # the re-probe failed
            st["off_at"] = self._queries
            if failed:   # the history missed these queries' scores: start it over (their k-th best arrives with the next calls)
                st["hist"] = []
                st["queries"] = 0
        return 0   # (a speculative call says nothing about the lists)

    def _demote(self, what: str):
        """int8 -> bf16 levels ("i8") or filter -> fp32 kernels ("filter"), until the re-probe.  A demotion within one interval
        of the previous re-probe of the same kind is a failed probe: the next one waits four times as long."""
        if what == "i8":
            self._i8_off = True
        else:
            self._filter_off = True
        if self._demoted[what] is None:
            probed_at = self._probed_at[what]
            if probed_at is not None and self._queries - probed_at < self._reprobe_after[what]:
                self._reprobe_after[what] = min(self._reprobe_after[what] * 4, 1 << 40)   # the probe failed
            else:
                self._reprobe_after[what] = self.REPROBE_QUERIES
            self._demoted[what] = self._queries

    def _maybe_reprobe(self, B: int):
        """Before a dispatch: lift a demotion whose interval has passed (filter first: it is the larger loss), on a call small
        enough that a failed probe is cheap."""
        if B > self.REPROBE_MAX_BATCH:
            return
        for what in ("filter", "i8"):
            at = self._demoted[what]
            if at is not None and self._queries - at >= self._reprobe_after[what]:
                if what == "i8":
                    self._i8_off = False
                    self._seen_i8 = [0, 0]
                else:
                    self._filter_off = False
                    self._seen_bf16 = [0, 0]
                self._demoted[what] = None
                self._probed_at[what] = self._queries
                return

    def _spec_state(self, k: int) -> dict:
        st = self._spec.get(k)
        if st is None:
            st = self._spec[k] = {"hist": [], "queries": 0, "off_at": None, "after": self.REPROBE_QUERIES, "cand": None,
                                  "failed": 0, "used": 0}
        return st

    def _prior_for(self, B: int, k: int, small: bool = False):
        """The speculative first bound for a call of B queries, or None (not warm yet, withdrawn, switched off, capturing).
        small: the single-launch kernel's call (any batch size it takes)."""
        if not self.spec_enabled or (B < self.SPEC_MIN_BATCH and not small) or getattr(self.ops, "set_filter_prior", None) is None:
            return None
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # a captured launch would carry TODAY's prior by value into every replay, and the statistics that could withdraw it
            # are not read under capture: captured calls keep their bound pass (a proof, whatever queries the replays bring)
            return None
        st = self._spec_state(k)
        if st["off_at"] is not None:
            if self._queries - st["off_at"] < st["after"] or B > self.REPROBE_MAX_BATCH:
                return None
            st["off_at"] = None          # re-probe: a miss now quadruples the interval (see _poll_overflow)
            st["probed_at"] = self._queries
        if len(st["hist"]) < self.SPEC_WARM_CALLS or st["queries"] < self.SPEC_WARM_QUERIES:
            return None
        lo = min(h[0] for h in st["hist"])
        hi = max(h[1] for h in st["hist"])
        return lo - max(self.SPEC_MARGIN * (hi - lo), self.SPEC_MIN_MARGIN)

    def _note_overflow(self, over, B: int, had_i8: bool, stats=None, k: int = 0):
        """After a filtered call: its overflow count travels to a pinned host word behind an event (no wait) and is
        judged by _poll_overflow at a later call."""
        if not over.is_cuda:  # (the CPU tests' oracle shim)
            self._overflowed += int(over)
        elif self._pending is None and not torch.cuda.is_current_stream_capturing():
            # a call of a few queries takes ~60 us and the copy + event below ~2 us of its stream: once the bank's dispatch
            # has settled (a prior in use, nothing withdrawn or demoted) such calls report every fourth time only
            self._notes += 1
            if B <= 64 and (self._notes & 3) and self.last_prior is not None:
                return
            if self._host_word is None:
                self._host_word = torch.zeros(33, dtype=torch.int32).pin_memory()   # [0] overflow, [1:33] the call's statistics
                self._event = torch.cuda.Event()
            if stats is not None and stats.numel() >= 32:
                # ONE copy: the statistics words carry the call's final overflow count themselves ([20])
                self._host_word[1:33].copy_(stats[:32], non_blocking=True)
                self._host_word[0] = -1            # (host store: "read it from word 20")
            else:
                self._host_word[:1].copy_(over, non_blocking=True)
                self._host_word[1:].zero_()
                if stats is not None:
                    self._host_word[1:1 + stats.numel()].copy_(stats, non_blocking=True)
            self._event.record()
            self._pending = (self._host_word, self._event, B, had_i8, k)

    def _cap_i8(self):
        """Before a filtered call: cap this thread's int8 levels for THIS bank (ops.set_max_i8_levels; the caller resets it
        to -1 afterwards).  The bank's int8 error row is read once per bank version (one synchronisation; never while a
        HIP graph is being captured -- an unjudged bank then stays on bf16).  Returns (the setter or None, int8 allowed)."""
        cap = getattr(self.ops, "set_max_i8_levels", None)
        if cap is None:
            return None, False
        if self._i8_ok is None and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            # (the copy has two scales, csrc/filter_common.h: the NORMAL granules' error decides, unless more than a quarter
            # of the granules are HEAVY ones with a useless bound -- a bank of heavy-tailed rows throughout)
            c = self.ops.int8_copy_classes(self._bf16, self.keys_normalized.shape[0])
            many_heavy = 4 * c["heavy_granules"] > c["granules"] and c["err_heavy"] > self.I8_MAX_ERR
            self._i8_ok = c["err"] <= self.I8_MAX_ERR and not many_heavy
            self.i8_classes = c
        allowed = bool(self._i8_ok) and not self._i8_off
        cap(-1 if allowed else 0)
        return cap, allowed

    def sharded_speculates(self, B: int, k: int, plan_n: int, n_shards: int) -> bool:
        """Would topk(..., exchange, plan_n, prior) of B queries skip its bound pass and exchange 0?  Taken from what every
        rank shares (B, k, the padded width, plan_n, the shard count): the same answer on every rank."""
        ops = self.ops
        if self._width is None or getattr(ops, "set_filter_prior", None) is None or getattr(ops, "sharded_speculates", None) is None:
            return False
        fhelps = getattr(ops, "filter_helps", None)
        if fhelps is None or B > self.MAX_FILTERED_BATCH or not fhelps(B, plan_n, self._width, k):
            return False
        return bool(ops.sharded_speculates(B, plan_n, self._width, k, n_shards))

    def topk(self, q: torch.Tensor, k: int, idx_base: int = 0, exchange=None, plan_n: int = 0, prior=None):
        """exchange / plan_n: this index holds one shard of a row-sharded bank (ShardedToyGraphBase): the filtered path
        sharpens its per-query bounds across the shards through `exchange` (kernels.topk_cosine_filtered), and every
        decision that changes which collectives run is taken from plan_n -- the largest shard's size -- so that all
        ranks take it alike.  prior (with exchange only): the speculative first bound the owner of the sharded bank chose for
        THIS call on every rank alike (it removes the bound pass and exchange 0; the owner proves the merged rows)."""
        ops, kn = self.ops, self.keys_normalized
        if q.shape[1] != self.dim:
            raise ValueError(f"KeyIndex.topk: queries of {q.shape[1]} columns against a bank of {self.dim}")
        if self._width is None:   # wider than every fused kernel: exact score slabs, this shard's own top-k (no exchange needed)
            return ops.topk_cosine(q, kn, k, idx_base=idx_base)
        if self._width != self.dim:
            q = ops.pad_cols(q, self._width)
        B, D = q.shape
        if self._collapsed is None:
            self._judge_duplicates()
        if self._collapsed:
            inner, group_ptr, members = self._collapsed
            U = inner.keys_normalized.shape[0]
            if exchange is None:
                su, iu = inner.topk(q, min(k, U))
                return ops.topk_expand_groups(su, iu, group_ptr, members, k, idx_base=idx_base)
            if U >= k:
                # one shard of a row-sharded bank: the unique rows take part in the exchanges (plan_n = the largest number of
                # searched rows over the shards -- ShardedToyGraphBase asks search_rows() of every rank); the shard's list
                # (padded with -inf / INT64_MAX where nothing can reach the global top-k any more) is expanded to bank rows
                su, iu = inner.topk(q, k, 0, exchange, plan_n, prior)
                return ops.topk_expand_groups(su, iu, group_ptr, members, k, idx_base=idx_base)
            raise RuntimeError("KeyIndex: a collapsed shard of fewer unique rows than k cannot take part in the exchanges; "
                               "ShardedToyGraphBase keeps such a shard uncollapsed (search_rows(min_unique=k))")
        fhelps = getattr(ops, "filter_helps", None)
        if exchange is not None:
            if fhelps is not None and fhelps(B, max(plan_n, kn.shape[0]), D, k) and B <= self.MAX_FILTERED_BATCH:
                if self._bf16 is None:
                    self._bf16 = ops.keys_to_bf16(kn)
                cap, _ = self._cap_i8()   # (per shard: which kernel a level runs on does not change the exchanges)
                if prior is not None:
                    ops.set_filter_prior(prior)
                try:
                    if getattr(ops, "FILTER_STATS", False):
                        s, i, self.last_over, self.last_stats = ops.topk_cosine_filtered(
                            q, kn, self._bf16, k, idx_base=idx_base, exchange=exchange, plan_n=plan_n, return_stats=True)
                    else:
                        s, i, self.last_over = ops.topk_cosine_filtered(q, kn, self._bf16, k, idx_base=idx_base,
                                                                        exchange=exchange, plan_n=plan_n)
                finally:
                    if cap is not None:
                        cap(-1)
                    if prior is not None:
                        ops.set_filter_prior(None)
                self.last_prior = prior
                return s, i
            fhelps = None  # (fp32 kernels: the shard's own exact top-k, no exchange needed)
        self._poll_overflow()
        self._maybe_reprobe(B)
        self._queries += B
        fused = getattr(ops, "fused_helps", None)
        if fused is not None and not self._filter_off and fused(B, kn.shape[0], D, k):  # small bank: every phase in one launch
            if self._bf16 is None:
                self._bf16 = ops.keys_to_bf16(kn)
            return ops.topk_cosine_fused(q, kn, self._bf16, k, idx_base=idx_base)
        small = getattr(ops, "small_helps", None)
        if (small is not None and not self._filter_off and small(B, kn.shape[0], D, k)
                and getattr(ops, "small_state_ready", lambda _d: True)(q.device)):
            # a handful of queries against a large bank: every phase of the filtered call in ONE launch (csrc/topk_small.hip)
            if self._bf16 is None:
                self._bf16 = ops.keys_to_bf16(kn)
            cap, had_i8 = self._cap_i8()
            stats = None
            prior = self._prior_for(B, k, small=True)
            if prior is not None:
                ops.set_filter_prior(prior)
            try:
                if getattr(ops, "FILTER_STATS", False):
                    s, i, over, stats = ops.topk_cosine_small(q, kn, self._bf16, k, idx_base=idx_base, return_stats=True)
                else:
                    s, i, over = ops.topk_cosine_small(q, kn, self._bf16, k, idx_base=idx_base)
            finally:
                if cap is not None:
                    cap(-1)
                if prior is not None:
                    ops.set_filter_prior(None)
            self.last_prior = prior
            self._note_overflow(over, B, had_i8 and D in (128, 256), stats, k)
            return s, i
        if fhelps is not None and not self._filter_off and fhelps(B, kn.shape[0], D, k):
            if self._bf16 is None:
                self._bf16 = ops.keys_to_bf16(kn)
            # (the packed fp32 copy is not made for this path: its first bound comes from the bf16 copy itself)
            if B > self.MAX_FILTERED_BATCH:
                # the candidate lists take 8 KiB per query: slabs of 256 k queries keep the workspace at 2 GiB (the edge
                # flavour asks for all 4 M nodes at once) at the throughput of one big call
                outs = [self.topk(q[b0:b0 + self.MAX_FILTERED_BATCH], k, idx_base)
                        for b0 in range(0, B, self.MAX_FILTERED_BATCH)]
                return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
            cap, had_i8 = self._cap_i8()
            had_i8 = had_i8 and ops.filtered_i8_levels(B, kn.shape[0], D, k) > 0   # (did THIS call have int8 levels?)
            stats = None
            prior = self._prior_for(B, k)
            if prior is not None:
                ops.set_filter_prior(prior)
            try:
                if getattr(ops, "FILTER_STATS", False):   # this call's own statistics words, handed on explicitly
                    s, i, over, stats = ops.topk_cosine_filtered(q, kn, self._bf16, k, idx_base=idx_base,
                                                                 keys_packed=self._packed, return_stats=True)
                else:
                    s, i, over = ops.topk_cosine_filtered(q, kn, self._bf16, k, idx_base=idx_base, keys_packed=self._packed)
            finally:
                if cap is not None:
                    cap(-1)
                if prior is not None:
                    ops.set_filter_prior(None)
            self.last_prior = prior
            self._note_overflow(over, B, had_i8, stats, k)
            self.last_stats = stats   # (diagnostic: bench.py / tools read the levels' candidate counts of the last call)
            return s, i
        helps = getattr(ops, "packed_keys_help", None)
        if helps is not None and helps(B, D, k):
            if self._packed is None:
                self._packed = ops.pack_keys(kn)
            return ops.topk_cosine(q, kn, k, idx_base=idx_base, keys_packed=self._packed)
        return ops.topk_cosine(q, kn, k, idx_base=idx_base)
