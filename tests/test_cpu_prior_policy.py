"""The host-side policy of the speculative first bound (ragraph_amd/kernels_index.py: KeyIndex._prior_for / _judge_prior) on
synthetic statistics words -- no GPU: when the prior appears, what it is, and what withdraws it."""
import struct

import torch

from ragraph_amd.kernels_index import KeyIndex

MAGIC = 0x52414753


def f2ord(x):
    b = struct.unpack("<i", struct.pack("<f", x))[0]
    return b if b >= 0 else b ^ 0x7FFFFFFF


class Ops:
    """Just enough of ragraph_amd.kernels for the policy code."""
    FILTER_STATS = True

    def __init__(self):
        self.prior = None

    def set_filter_prior(self, p):
        self.prior = p


def words(spec, failed, lo, hi, cand=100.0, B=512):
    w = [0] * 32
    w[0], w[1] = MAGIC, 1
    w[2], w[5] = int(cand * 8), 8          # sampled candidates / sampled queries of level 0
    w[14], w[16], w[17], w[18], w[19] = B, spec, failed, f2ord(lo), f2ord(hi)
    return w


def index():
    idx = KeyIndex(torch.zeros(4, 64), ops=Ops(), dedup=False)
    idx._queries = 0
    return idx


def test_prior_appears_after_two_calls_and_sits_below_everything_seen():
    idx = index()
    assert idx._prior_for(512, 10) is None                                  # nothing seen yet
    assert idx._judge_prior(10, 512, words(0, 0, 0.250, 0.280), 0) == 0
    assert idx._prior_for(512, 10) is None                                  # one call is not enough
    idx._judge_prior(10, 512, words(0, 0, 0.255, 0.290), 0)
    p = idx._prior_for(512, 10)
    assert abs(p - (0.250 - 0.5 * (0.290 - 0.250))) < 1e-6                   # lowest seen - half the spread
    assert idx._prior_for(16, 10) is None                                   # the single-launch kernel's batch sizes: never
    assert idx._prior_for(512, 5) is None                                   # another k has its own history
    idx2 = index()
    for _ in range(3):
        idx2._judge_prior(10, 512, words(0, 0, 0.3000, 0.3001), 0)
    assert abs(idx2._prior_for(512, 10) - (0.3000 - KeyIndex.SPEC_MIN_MARGIN)) < 1e-6   # a narrow spread: the minimum margin
    idx2.spec_enabled = False
    assert idx2._prior_for(512, 10) is None


def test_a_miss_withdraws_the_prior_and_is_not_blamed_on_the_lists():
    idx = index()
    for _ in range(2):
        idx._judge_prior(10, 512, words(0, 0, 0.25, 0.28), 0)
    assert idx._prior_for(512, 10) is not None
    left = idx._judge_prior(10, 512, words(1, 7, 0.26, 0.28), 7)             # a speculative call with 7 misses, 7 "overflowed"
    assert left == 0                                                        # nothing for the int8 / filter demotion rules
    st = idx._spec[10]
    assert st["failed"] == 7 and st["off_at"] is not None and st["hist"] == []
    assert idx._prior_for(512, 10) is None
    # the re-probe interval passes (and the history refills): speculation is tried again on a small call
    for _ in range(2):
        idx._judge_prior(10, 512, words(0, 0, 0.20, 0.28), 0)
    idx._queries += KeyIndex.REPROBE_QUERIES
    assert idx._prior_for(100_000, 10) is None                               # not on a call this large
    assert idx._prior_for(512, 10) is not None
    # a miss right behind the re-probe quadruples the interval
    idx._judge_prior(10, 512, words(1, 1, 0.21, 0.28), 1)
    assert st["after"] == 4 * KeyIndex.REPROBE_QUERIES


def test_loose_priors_and_overflowing_lists_withdraw_it_too():
    idx = index()
    for _ in range(2):
        idx._judge_prior(10, 512, words(0, 0, 0.25, 0.28, cand=100.0), 0)
    assert idx._judge_prior(10, 512, words(1, 0, 0.25, 0.28, cand=140.0), 0) == 0
    assert idx._spec[10]["off_at"] is None                                   # 1.4 x the bound pass's candidates: kept
    idx._judge_prior(10, 512, words(1, 0, 0.25, 0.28, cand=400.0), 0)
    assert idx._spec[10]["off_at"] is not None                               # 4 x: the prior is too loose to pay
    idx = index()
    for _ in range(2):
        idx._judge_prior(10, 512, words(0, 0, 0.25, 0.28), 0)
    assert idx._judge_prior(10, 512, words(1, 0, 0.25, 0.28), 30) == 0       # lists overflowed UNDER the prior: its fault
    assert idx._spec[10]["off_at"] is not None
    idx = index()
    assert idx._judge_prior(10, 512, words(0, 0, 0.25, 0.28), 30) == 30      # ... with a bound pass: the lists' own
    assert idx._spec[10]["hist"] == []                                       # and such a call is no ground for a prior


def test_int8_gate_judges_the_normal_granules_of_a_two_scale_copy():
    """KeyIndex._cap_i8 on synthetic class records (kernels.int8_copy_classes): the NORMAL granules' measured error decides;
    a few heavy granules with a useless bound do not matter, a bank that is heavy throughout leaves int8, and the per-thread cap
    is what the call sees."""
    class CapOps(Ops):
        def __init__(self, rec):
            super().__init__()
            self.rec, self.caps = rec, []

        def int8_copy_classes(self, kb, n):
            return dict(self.rec)

        def set_max_i8_levels(self, n):
            self.caps.append(n)
            return -1

    def gate(rec):
        idx = KeyIndex(torch.zeros(4, 64), ops=CapOps(rec), dedup=False)
        idx._bf16 = torch.zeros(1, 64, dtype=torch.int16)
        cap, allowed = idx._cap_i8()
        return allowed, idx.ops.caps[-1], idx

    base = {"err": 0.0116, "scale": 0.0022, "max_abs": 0.35, "err_heavy": 0.0141, "scale_heavy": 0.0028, "cut": 0.28,
            "heavy_granules": 218, "granules": 1564}
    assert gate(base)[:2] == (True, -1)                                              # a Gaussian bank
    one_hot = dict(base, max_abs=1.0, err_heavy=0.040, scale_heavy=1 / 127, heavy_granules=11)
    allowed, cap, idx = gate(one_hot)
    assert (allowed, cap) == (True, -1) and idx.i8_classes["heavy_granules"] == 11   # a one-hot row costs its granules only
    assert gate(dict(one_hot, heavy_granules=400))[:2] == (False, 0)                 # heavy throughout: > 1/4 of the granules
    assert gate(dict(base, err=0.03))[:2] == (False, 0)                              # ordinary rows too coarse: as before
    allowed, cap, idx = gate(base)
    idx._i8_off = True                                                               # demoted by its calls: the cap follows
    assert idx._cap_i8()[1] is False and idx.ops.caps[-1] == 0


def test_group_prior_policy_is_a_function_of_the_pooled_numbers():
    """ragraph_amd.sharded.GroupPrior (round 6): two ranks that feed the same pooled words derive the same prior at the same
    call, withdraw it at the same call after a miss / a flood of candidates / overflowed lists, re-probe after the same number
    of calls and back off four-fold when the re-probe fails -- the number of exchanges of a sharded call can never differ
    between ranks."""
    from ragraph_amd.sharded import GroupPrior

    a, b = GroupPrior(), GroupPrior()
    k, B = 10, 1000
    feed = [  # (speculative?, misses, lo, hi, cand, lists_over) as the all_reduce leaves them on every rank
        (False, 0, 0.240, 0.280, 50.0, 0), (False, 0, 0.238, 0.279, 52.0, 0),
    ]
    for ga in (a, b):
        assert ga.prior_for(B, k) is None and ga.prior_for(5, k) is None
        for w in feed:
            assert ga.record(k, *w) is True
    pa, pb = a.prior_for(B, k), b.prior_for(B, k)
    assert pa == pb and abs(pa - (0.238 - 0.5 * (0.280 - 0.238))) < 1e-12
    assert a.prior_for(16, k) is None                                  # below the smallest speculative batch
    for ga in (a, b):
        assert ga.record(k, True, 0, 0.239, 0.281, 60.0, 0) is True     # a speculative call that stands
        assert ga.prior_for(B, k) is not None
        assert ga.record(k, True, 3, 0.250, 0.281, 60.0, 0) is False    # three rows missed: the call is repeated ...
        assert ga.prior_for(B, k) is None                               # ... and the prior withdrawn on every rank alike
    # the history restarts: two calls with a bound pass, but the withdrawal lasts REPROBE_CALLS calls
    for ga in (a, b):
        for _ in range(GroupPrior.REPROBE_CALLS - 1):
            ga.record(k, False, 0, 0.24, 0.28, 50.0, 0)
            assert ga.prior_for(B, k) is None
        ga.record(k, False, 0, 0.24, 0.28, 50.0, 0)
        assert ga.prior_for(B, k) is not None                           # the re-probe
        assert ga.record(k, True, 0, 0.24, 0.28, 500.0, 0) is True      # a flood of candidates: stands, but withdrawn again
        assert ga.prior_for(B, k) is None and ga._state(k)["after"] == 4 * GroupPrior.REPROBE_CALLS
    assert a.calls == b.calls and a.used == b.used == 3 and a.missed_calls == 1
    c = GroupPrior()
    c.forced = 0.3
    assert c.prior_for(3, k) == 0.3                                      # (tests force a prior whatever the history says)
    c.forced = None
    c.record(k, False, 0, 0.2, 0.3, 40.0, 2)                             # overflowed lists are no ground for a prior
    c.record(k, False, 0, 0.2, 0.3, 40.0, 0)
    assert c.prior_for(B, k) is None
