"""Two real processes on the real HIP library (-m gpu): the key-sharded and the query-sharded forward of RAGraph_node
under a world-size-2 process group, bit for bit against the single-process result.  The ranks are started by
tests/conftest.py at session start -- before this pytest process has made any GPU call (a process that has initialised
the GPU must not fork + exec on the GPU pool) -- and run beside the other tests; this test collects their reports."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_on_the_hip_library_match_single_process(dev, two_rank_job):
    assert two_rank_job is not None, "conftest did not start the two-rank job (no ROCm device at session start?)"
    procs, out = two_rank_job[0], two_rank_job[1]
    for p in procs:
        p.wait(timeout=900)
    reports = []
    for r in range(len(procs)):
        path = os.path.join(out, f"rank{r}.json")
        assert os.path.exists(path), f"rank {r} left no report (exit code {procs[r].returncode})"
        reports.append(json.load(open(path)))
    for rep in reports:
        assert rep["ok"], rep.get("error")
        assert rep["world"] == 2
        assert rep["filtered_path"]                          # the shard goes through the bf16-filtered sharded entry
        assert rep["key_shard_forward_equal"] and rep["key_shard_topk_equal"] and rep["query_shard_forward_equal"], rep
        assert rep["key_shard_topk_d64_equal"], rep
        assert rep["small_launch_equal"] and rep["small_launch_calls"] == 120, rep   # the single-launch kernel, two processes at once
        # (a workgroup whose bounded wait expired -- the other process held the CUs -- ratchets its bound from its own exact
        # scores instead of flooding the lists: nothing may end in the exact scan on this ordinary bank)
        assert rep["small_launch_overflowed"] == 0, rep
        assert rep["exchange_count"].get("0", 0) >= 2, rep   # phase-0 exchange: once per key-sharded retrieval with a bound pass
        # round 6: the group's speculative first bound on the real library -- learnt by the policy, then forced
        assert rep["spec_forwards_equal"] and rep["spec_calls_used"] >= 1 and rep["spec_reruns_auto"] == 0, rep
        assert rep["spec_phase0_during_auto"] < rep["spec_group_calls"], rep          # calls without phase 0 did happen
        f = rep["spec_forced"]
        assert all(f[t]["equal"] for t in ("low", "one", "high")), rep
        assert (f["low"]["reruns"], f["low"]["phase0"]) == (0, 0), rep               # stands: no repeat, no phase 0
        assert (f["one"]["reruns"], f["one"]["phase0"]) == (1, 1), rep               # ONE row missed: every rank repeated the call
        assert (f["high"]["reruns"], f["high"]["phase0"]) == (1, 1), rep
    assert reports[0]["spec_one_row_owner"] != reports[1]["spec_one_row_owner"]       # the missed row had exactly one owner


def test_four_ranks_hybrid_layout(dev, two_rank_job):
    """2 query groups x 2 key shards (HybridLayout) with four real processes on the real library: every rank's gathered
    [n, C] output equals the single-process forward bit for bit, and the exchanges stayed inside the key groups."""
    assert two_rank_job is not None and len(two_rank_job) >= 6, "conftest did not start the hybrid job"
    procs, out = two_rank_job[3], two_rank_job[4]
    for p in procs:
        p.wait(timeout=900)
    reports = []
    for r in range(len(procs)):
        path = os.path.join(out, f"rank{r}.json")
        assert os.path.exists(path), f"rank {r} left no report (exit code {procs[r].returncode})"
        reports.append(json.load(open(path)))
    for r, rep in enumerate(reports):
        assert rep["ok"], rep.get("error")
        assert rep["world"] == 4 and rep["q_s"] == [r // 2, r % 2] and rep["layout"].startswith("hybrid 2x2")
        assert rep["hybrid_forward_equal"], rep
        assert rep["exchange_count"].get("0", 0) >= 1, rep
        assert rep["forced_one_forward_equal"], rep
    for g in (0, 2):   # the two ranks of a key group went through the same exchanges
        assert reports[g]["exchange_count"] == reports[g + 1]["exchange_count"]
        assert reports[g]["forced_one_reruns"] == reports[g + 1]["forced_one_reruns"]   # a repeat is the key group's decision
    # the one row that missed the forced prior belongs to ONE query group: that group repeated its retrieval, the other did not
    assert sorted(reports[g]["forced_one_reruns"] for g in (0, 2)) == [0, 1]
