#!/usr/bin/env python3
"""bench.py -- RAGraph retrieve-and-propagate hot path on MI355X.

Workload (BASELINE.json configs[1]): RAGraph_node forward on a synthetic 100k-node graph (F=128, mean degree ~10)
against a 1M-key x 256-d bank, k=10, C=3.  One step = one full forward: GCN encode -> exact cosine top-k retrieval of
every node against the bank -> winners' value-sum / label-mean -> 3-hop propagation -> fusion + decoder + softmax-mix.
value = retrieved queries (= nodes) per second, whole job; the timed region is exactly `--steps` forwards.

N > 1 (one rank per GPU; `python bench.py --gpus N` starts the ranks itself when no launcher did): strong scaling on the
metric's own 1M-key bank.  Default layout = north_star's: the KEY BANK is row-sharded, every rank scores all queries
against its shard, and between the phases of that call (first bound, every filter level) the ranks pool their bounds: an
all_gather of each rank's best m = 2 ceil(k/G) lower bounds per query + the k-th largest of the union
(ragraph_theta_sharpen_f32), so a shard filters with (nearly) the global threshold.  Since round 6 the call runs under the
GROUP's speculative first bound once two calls have reported where the merged k-th best scores lie (ragraph_amd.sharded.GroupPrior:
no bound pass, no phase-0 exchange; the rows' owners prove the merged lists, one 5-float all_reduce makes the verdict the group's,
a miss anywhere repeats the call without the prior on every rank) -- `first_bound.group` and every layout's `first_bound` count the
calls, the speculative ones and the repeats.  The tail is query-sharded: ONE
all_to_all carries the per-shard lists of a rank's rows to it, the canonical merge, value gathers, last hop and decoder
run on B/G rows, and one all_gather of the [n, C] outputs completes the step (ragraph_amd/sharded.py,
RAGraph._forward_key_shard); values / labels are replicated (1 GB), the cheap GNN part runs on every rank.  The other
layout (bank replicated, QUERY batch split, no data-path collective) is timed right after and reported as
`query_sharded`; `--shard queries` makes it the headline instead.  `--backend gloo` runs the same job over gloo (device
tensors staged through the host: two ranks on ONE GPU, tests/test_gpu_two_rank.py).

The steps cycle `--batches` (8) DISTINCT feature tensors (seeds 4321, 4322, ...: same distribution) through warm-up and the
timed region, so the filtered call's learnt first bound meets queries it has not seen; `first_bound` reports the prior each
timed step ran under and the queries its verify launch had to scan (`repeated_batch`: round 5's one-batch replay, secondary).

After the timed region (outside it) the LAST timed batch's retrieval is CHECKED: 1024 evenly spaced queries are re-scored on
the exact fp32 kernel (same indices and score bits required) and 4 of them against the CPU oracle (`verified`).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` for the dominant kernel (the bf16 MFMA
filter of the exact top-k) and, at N = 1, next to the timed region: `retrieval_small_batch` (the reference's real batch
sizes, B = 1 / 16 / 256 / 512 / 4096, HBM-bound up to a few hundred queries), `exact_fp32` (the same step on the fp32
MFMA kernels alone), `gnn_fwd` (the metric's second half: GNN-forward nodes/s with SURVEY 8(d)'s byte model, the counter
bytes of the same kernels, the CPU port's rate and a graph with structure), `configs` (c1, c3, few-shot, the c5-shaped
single-GPU leg -- each with its rate and roofline fraction), `finetune_step` (forward + loss + backward + Adam of the node
and edge flavours beside the torch-CPU restatement), `memory` (bytes of every bank image), and `cpu_baseline` (the
torch-CPU port of the reference's op chain, oracle/ref_torch.py: 1 warm-up + 3 repetitions of a 1024-query slab, median,
plus the 'fair' pre-normalised-bank row).  The blocks live in tools/bench_blocks.py.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: ~2.5 PF dense = 1024 flop/clk/SIMD x 1024 SIMDs x 2.4 GHz
INT8_MFMA_PEAK_TOPS = 5033.2    # same guide, Matrix cores: I8 "the cycles of the BF16 form at 2x the K, so 2x BF16 per clock"
HBM_PEAK_GBS = 8000.0           # spec; ~6300 achievable
VERIFY_ROWS = 1024              # rows of the timed path re-scored on the exact fp32 kernel after the timed region
TRAFFIC_JSON = next((p for p in (os.path.join(ROOT, "profiles", f"r{r}_pmc_traffic.json") for r in (6, 5, 4)) if os.path.exists(p)),
                    os.path.join(ROOT, "profiles", "r5_pmc_traffic.json"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nodes", type=int, default=100_000)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--bank", type=int, default=1_000_000,
                    help="key-bank rows (BASELINE configs[1]: 1 000 000; configs[3] = 8 000 000 with --gpus 8)")
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--batches", type=int, default=8,
                    help="distinct query batches (feature tensors of different seeds, same distribution) cycled through the "
                         "warm-up and the timed steps; 1 = round 5's repeated batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` and `finetune_step` blocks (c1, c3, few-shot, c5)")
    ap.add_argument("--cpu-slab", type=int, default=1024, help="queries per CPU-baseline repetition (one slab)")
    ap.add_argument("--shard", choices=("keys", "queries", "hybrid"), default="keys",
                    help="N > 1: row-shard the key bank (north_star's layout; default), split the query batch over the GPUs "
                         "with the bank replicated, or both (--key-shards S key shards x N / S query groups)")
    ap.add_argument("--key-shards", type=int, default=2, help="--shard hybrid: key shards per query group (2 x 4 at N = 8)")
    ap.add_argument("--dist-timeout", type=float, default=120.0,
                    help="N > 1: seconds a collective (and the watchdog over every rank's heartbeat) waits before the job "
                         "exits non-zero with every rank's last phase")
    ap.add_argument("--emulate-rank-of", type=int, default=0, metavar="G",
                    help="single process: time what rank 0 of a G-GPU job would compute (collectives replaced by their "
                         "local part); an estimate of the per-rank step for DESIGN.md, never the bench line of a real "
                         "multi-GPU run")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend for N > 1: nccl (= RCCL over xGMI) or gloo (device tensors staged through "
                         "the host; lets several ranks share one GPU for a functional check)")
    ap.add_argument("--exact-fp32", action="store_true",
                    help="retrieve with the fp32 MFMA kernel only (no bf16 filter) in the timed region")
    return ap.parse_args()


class EventTimer:
    """Wraps a kernels.* entry so each call is bracketed by events on the stream it launches on (torch's current
    stream -- the C ABI is handed exactly that stream)."""

    def __init__(self, module, name):
        self.module, self.name, self.orig = module, name, getattr(module, name)
        self.events = []
        self.enabled = False
        self.last = None   # the wrapped call's last result (the filtered call: its statistics words ride as a fourth item)
        setattr(module, name, self)

    def __call__(self, *a, **kw):
        if not self.enabled:
            return self.orig(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = self.orig(*a, **kw)
        e1.record()
        self.events.append((e0, e1))
        self.last = out
        return out

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.events) / max(len(self.events), 1)

    def reset(self):
        self.events = []


def _timeout(args):
    import datetime

    return datetime.timedelta(seconds=float(args.dist_timeout))


class Heartbeat:
    """N > 1: every rank keeps `<dir>/rank<r>.txt` = "<unix time> <phase>" up to date (bench phases, every collective and
    every exchange of ragraph_amd.sharded); a watchdog thread per rank exits the process non-zero -- after printing EVERY
    rank's last line to stderr -- when its own phase has not changed for `timeout` seconds: a hung exchange names the phase
    each rank is waiting in instead of sitting until the driver's limit.  (os._exit from the thread: nothing is re-executed,
    the launcher sees the non-zero code and ends the other ranks.)"""

    def __init__(self, rank, world, timeout):
        import tempfile
        import threading

        self.rank, self.world, self.timeout = rank, world, float(timeout)
        self.dir = os.environ.get("RAGRAPH_HEARTBEAT_DIR") or os.path.join(
            tempfile.gettempdir(), f"ragraph_bench_{os.environ.get('MASTER_PORT', '0')}")
        os.makedirs(self.dir, exist_ok=True)
        self.path = os.path.join(self.dir, f"rank{rank}.txt")
        self.last = time.time()
        self.phase = "start"
        self.done = False
        self.headline = None   # the finished headline record, once there is one (see _watch)
        self("start")
        self.thread = threading.Thread(target=self._watch, daemon=True)
        self.thread.start()

    def __call__(self, phase: str):
        self.last, self.phase = time.time(), phase
        try:
            with open(self.path, "w") as f:
                f.write(f"{self.last:.3f} {phase}\n")
        except OSError:
            pass

    def report(self):
        out = {}
        for r in range(self.world):
            try:
                t, ph = open(os.path.join(self.dir, f"rank{r}.txt")).read().strip().split(" ", 1)
                out[f"rank{r}"] = f"{ph} ({time.time() - float(t):.0f} s ago)"
            except (OSError, ValueError):
                out[f"rank{r}"] = "no heartbeat file"
        return out

    def _watch(self):
        while not self.done:
            time.sleep(min(5.0, self.timeout / 4))
            if not self.done and time.time() - self.last > self.timeout:
                print(json.dumps({"bench_watchdog": f"rank {self.rank}: no progress for {self.timeout:.0f} s in phase "
                                                    f"'{self.phase}'", "ranks": self.report()}), file=sys.stderr, flush=True)
                if self.phase.startswith("layout ") and self.headline is not None:
                    # the HEADLINE layout has been timed and verified; only an extra layout behind it stopped: every rank
                    # leaves with 0 and rank 0 prints the line it has (the stopped layout named in it)
                    if self.rank == 0:
                        line = dict(self.headline, other_layouts_error=f"stopped in phase '{self.phase}' ({self.timeout:.0f} s)")
                        print(json.dumps(line), flush=True)
                    os._exit(0)
                os._exit(124)

    def stop(self):
        self.done = True


def preflight(rank, world, dev, backend, beat):
    """Before anything large is built: does the process group see every rank?  all_reduce of ones, all_gather of every rank's
    device index; rank 0 prints the record to STDERR (stdout stays the one JSON line) and every rank raises when a rank is
    missing."""
    from ragraph_amd import sharded as SH

    beat("preflight all_reduce")
    ones = torch.ones(1, device=dev, dtype=torch.int64)
    SH.all_reduce(ones, dist.ReduceOp.SUM)
    beat("preflight all_gather")
    mine = torch.tensor([rank, dev.index if dev.index is not None else 0], device=dev, dtype=torch.int64)
    seen = torch.empty((world, 2), device=dev, dtype=torch.int64)
    SH.all_gather_into(seen, mine.reshape(1, 2))
    rec = {"world_size": dist.get_world_size(), "all_reduce_of_ones": int(ones.item()), "backend": dist.get_backend(),
           "rank_devices": {str(int(r)): int(d) for r, d in seen.tolist()}, "device_count": torch.cuda.device_count(),
           "device_name": torch.cuda.get_device_name(dev)}
    if rank == 0:
        print(json.dumps({"ranks_seen": rec}), file=sys.stderr, flush=True)
    if rec["all_reduce_of_ones"] != world or sorted(int(r) for r in rec["rank_devices"]) != list(range(world)):
        raise SystemExit(f"bench.py: the process group does not see every rank: {rec}")
    return rec


def build_workload(args, dev, rank, world, shard, force_dist=False):
    from ragraph_amd import kernels as K
    from ragraph_amd.data import synthetic_bank, synthetic_big_graph
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph
    from ragraph_amd.sharded import QueryShard, ShardedToyGraphBase, shard_bounds

    torch.manual_seed(0)
    pre = PrePrompt(args.feat, args.dim, "prelu", 1, 0.3).to(dev)
    model = RAGraph(pre, None, args.feat, args.classes, args.dim, finetune=True, device=dev)
    model.toy_graph_base.retrieve_num = args.k
    model.eval()
    ei = synthetic_big_graph(args.nodes, 10, seed=8, device=dev)
    adj = CSRGraph.from_edge_index_sym_normalized(ei, args.nodes)
    feats = torch.randn(args.nodes, args.feat, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
    Kb, Vb, Lb = synthetic_bank(args.bank, args.dim, args.classes, device=dev)
    Kb = K.normalize_rows(Kb)  # stored unit-norm, as the reference stores keys (ToyGraphBase.py:109)
    emu = args.emulate_rank_of
    G = emu if emu > 1 else world
    if G > 1 or force_dist:
        if shard == "hybrid":
            from ragraph_amd.sharded import HybridLayout

            S = args.key_shards
            if emu > 1:   # rank 0 of emu = Q x S ranks: the exchanges of S shards, a Q-th of the queries
                class _Slice:
                    world, rank, collective = emu // S, 0, False

                    def bounds(self, B):
                        return shard_bounds(B, emu // S, 0)

                    def gather_rows(self, local, B):
                        return local
                model.query_shard = _Slice()
                lo, hi = shard_bounds(args.bank, S, 0)
                model.toy_graph_base = ShardedToyGraphBase(Kb[lo:hi].contiguous(), Vb, Lb, lo, args.k, values_replicated=True,
                                                           emulate_world=S)
            else:
                layout = HybridLayout(S, timeout=_timeout(args))
                lo, hi = layout.key_rows(args.bank)
                model.toy_graph_base = ShardedToyGraphBase(Kb[lo:hi].contiguous(), Vb, Lb, lo, args.k, group=layout.key_group,
                                                           force_collectives=force_dist, values_replicated=True)
                model.query_shard = layout.query_shard()
                model.layout_name = layout.name
            n_local = hi - lo
        elif shard == "queries":
            if emu > 1:
                class _Slice:  # rank 0 of `emu`, no process group
                    world, rank, collective = emu, 0, False

                    def bounds(self, B):
                        return shard_bounds(B, emu, 0)

                    def gather_rows(self, local, B):
                        return local
                model.query_shard = _Slice()
            else:
                model.query_shard = QueryShard(force_collectives=force_dist)
            model.toy_graph_base.set_resources(Kb, Vb, Lb)
            n_local = args.bank
        else:
            r = 0 if emu > 1 else rank
            lo, hi = shard_bounds(args.bank, G, r)
            # keys row-sharded; values / labels replicated (1 GB of 288 GB): the winners' sums are local gathers
            model.toy_graph_base = ShardedToyGraphBase(Kb[lo:hi].contiguous(), Vb, Lb, lo, args.k,
                                                       force_collectives=force_dist, values_replicated=True,
                                                       emulate_world=emu if emu > 1 else 0)
            n_local = hi - lo
    else:
        model.toy_graph_base.set_resources(Kb, Vb, Lb)
        _ = model.toy_graph_base.keys_normalized
        n_local = args.bank
    if emu > 1 and shard in ("keys", "hybrid") and os.environ.get("RAGRAPH_SPEC", "1") != "0":
        # An emulated rank has ONE shard: what its group would have learnt -- the prior comes from the MERGED k-th best scores of
        # the whole bank, pooled over the ranks -- cannot be learnt from that shard's lists.  It is computed here once, outside
        # every timing, from the whole bank (the policy's own rule over one batch), and the emulated rank runs under it;
        # sharded.py skips the verdict in emulation (timing only: an emulated rank's lists are not the global top-k anyway).
        from ragraph_amd.sharded import GroupPrior
        with torch.no_grad():
            h = model.pretrain_model.inference(feats, adj)
            s_all, _ = K.KeyIndex(Kb).topk(h, args.k)
        kth = s_all[:, args.k - 1]
        lo_, hi_ = float(kth.min()), float(kth.max())
        model.toy_graph_base.prior.forced = lo_ - max(GroupPrior.MARGIN * (hi_ - lo_), GroupPrior.MIN_MARGIN)
        del s_all, h
    del Kb
    torch.cuda.synchronize()
    return model, feats, adj, n_local


def gnn_only_rate(model, feats, adj, steps):
    from ragraph_amd.ragraph_utils import Propagation

    @torch.no_grad()
    def run():
        h = model.pretrain_model.inference(feats, adj)
        return Propagation.aggregate_k_hop_features(adj, h, model.query_graph_hop)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    return feats.shape[0] * steps / (time.perf_counter() - t0)


def small_batch_rates(tgb, dim, k, dev):
    """Retrieval alone at the reference's real batch sizes through the product dispatch (KeyIndex): graph
    classification sends ONE query per forward, RAGraph_node a few hundred, the edge flavour slabs of 4096.  Up to a few
    hundred queries a call is bound by one pass over a compressed bank copy: `streamed_GB` = the bytes the call streams from
    HBM (the prefix its bound pass reads on the bf16 copy + the filter pass, on the int8 copy where the schedule says so:
    D bytes per key instead of 2 D), `frac_hbm_peak` = that / time / 8 TB/s; `algorithmic_GBps` = SURVEY section 8(d)'s
    byte model (the fp32 bank once, 4 N D) / time, `algorithmic_frac_hbm_peak` that against the 8 TB/s spec (above 1: the
    call moves fewer bytes than the model's fp32 pass would)."""
    from ragraph_amd import kernels as K

    out = {}
    kn = tgb.keys_normalized
    index = tgb._index if tgb._index is not None else K.KeyIndex(kn)
    L = K.N.lib()
    for B in (1, 16, 256, 512, 4096):
        q = torch.randn(B, dim, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
        for _ in range(4):            # (the dispatch settles: overflow counts and the calls' statistics arrive one call late)
            index.topk(q, k)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            index.topk(q, k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        inner = index.search_index               # (the rows the kernels stream: the unique ones of a collapsed bank)
        n_keys = inner.keys_normalized.shape[0]
        filtered = K.filter_helps(B, n_keys, dim, k) and not inner._filter_off
        n_i8 = 0
        spec = False
        one_launch = filtered and K.small_helps(B, n_keys, dim, k)
        if one_launch:   # csrc/topk_small.hip: a bf16 prefix (bound pass) + one pass over the int8 (or bf16) copy, one launch
            cap, allowed = inner._cap_i8()
            i8 = ctypes.c_int(0)
            prefix = L.ragraph_topk_cosine_small_prefix_keys(B, n_keys, dim, ctypes.byref(i8))
            if cap is not None:
                cap(-1)
            n_i8 = i8.value
            spec = inner.last_prior is not None   # (a speculative first bound: no bound prefix is streamed)
            if spec:
                prefix = 0
            streamed = prefix * dim * 2 + n_keys * dim * (1 if n_i8 else 2) + B * dim * 4
            t_mfma = 2.0 * B * n_keys * dim / ((INT8_MFMA_PEAK_TOPS if n_i8 else BF16_MFMA_PEAK_TFLOPS) * 1e12)
        elif filtered:
            plan = (ctypes.c_int64 * 7)()
            L.ragraph_topk_cosine_filtered_plan(B, n_keys, dim, k, plan)
            cap, allowed = inner._cap_i8()       # the int8 levels THIS bank's calls run with (KeyIndex caps them per bank)
            n_i8 = L.ragraph_topk_cosine_filtered_i8_levels(B, n_keys, dim, k) if allowed else 0
            if cap is not None:
                cap(-1)
            ends = [0] + [int(plan[3 + l]) for l in range(int(plan[2]))]
            per_key = [dim * (1 if l >= int(plan[2]) - n_i8 else 2) for l in range(int(plan[2]))]   # int8 levels: D bytes per key
            spec = inner.last_prior is not None   # (a speculative first bound from the bank's own statistics: no bound pass)
            streamed = (0 if spec else int(plan[6]) * dim * 2) + sum((ends[l + 1] - ends[l]) * per_key[l] for l in range(int(plan[2]))) + B * dim * 4
            # the score matrix at the dense peak of the dtype each level runs on (int8 levels: 2x the bf16 peak)
            t_mfma = sum(2.0 * B * (ends[l + 1] - ends[l]) * dim /
                         ((INT8_MFMA_PEAK_TOPS if l >= int(plan[2]) - n_i8 else BF16_MFMA_PEAK_TFLOPS) * 1e12) for l in range(int(plan[2])))
        else:
            streamed = n_keys * dim * 4 + B * dim * 4
            t_mfma = 2.0 * B * n_keys * dim / (FP32_MFMA_PEAK_TFLOPS * 1e12)
        gbs = streamed / ms / 1e6
        flops = 2.0 * B * n_keys * dim
        rec = {"ms": round(ms, 4), "queries_per_s": round(B / ms * 1e3, 1),
               "path": (("one launch (" + ("speculative first bound: no bound pass; " if spec else "bound pass + ") + ("int8" if n_i8 else "bf16") +
                         " filter pass + exact rescoring + selection)" + (" + the (empty) sliced-scan launch for misses" if spec else ""))) if one_launch
               else (("bf16-filtered" + (f", last {n_i8} level(s) on int8" if n_i8 else "") +
                      (", speculative first bound (no bound pass; answers proven behind the last level)" if spec else ""))
                     if filtered else "fp32"),
               "streamed_GB": round(streamed / 1e9, 4), "GBps_streamed": round(gbs, 1),
               "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
               "algorithmic_GBps": round(n_keys * dim * 4 / ms / 1e6, 1),
               "algorithmic_frac_hbm_peak": round(n_keys * dim * 4 / ms / 1e6 / HBM_PEAK_GBS, 4),
               "TFLOPs": round(flops / ms / 1e9, 1)}
        # which roofline binds this batch: one pass over the streamed copy, or the score matrix on the matrix cores at the
        # peak of the dtype every level actually runs on
        t_hbm = streamed / (HBM_PEAK_GBS * 1e9)
        rec["bound"] = "hbm" if t_hbm >= t_mfma else "mfma"
        nlev = int(plan[2]) if (filtered and not one_launch) else 1
        rec["mfma_dtype"] = (("int8" if n_i8 >= nlev else "bf16 + int8") if n_i8 else "bf16") if (one_launch or filtered) else "f32"
        rec["frac_of_bound"] = round(max(t_hbm, t_mfma) / (ms * 1e-3), 4)
        out[f"B{B}"] = rec
    return out


def reference_bank_rates(args, dev, adj, feats, batches=(1, 500, 100_000), dedup=True, reps=None):
    """Retrieval against a bank shaped like the reference's OWN banks instead of SURVEY section 8(d)'s Gaussian one:
    args.bank rows made by the reference's recipe (ragraph_amd.bank_build.build_reference_recipe_bank: ToyGraphBase.py:91-119
    over synthetic resource graphs with the query graph's feature distribution, an encoder with a non-zero bias as after
    pre-training) -- three quarters of the rows are one vector (Augmentation.py:9-20 zeroes the augmented passes'
    features) and the sampled rows repeat (multinomial with replacement).  Queries = the bench graph's node embeddings by
    the same encoder.  Through the product dispatch (KeyIndex: exact duplicates collapsed, winners expanded in canonical
    order); 64 rows of the largest batch are checked against the fp32 kernel over every row (bits)."""
    from ragraph_amd import kernels as K
    from ragraph_amd.bank_build import build_reference_recipe_bank
    from ragraph_amd.preprompt import PrePrompt

    state = torch.random.get_rng_state()
    torch.manual_seed(1)
    pre = PrePrompt(args.feat, args.dim, "prelu", 1, 0.3).to(dev)
    t0 = time.perf_counter()
    with torch.no_grad():
        pre.gcn.convs[0].bias.normal_(0, 0.1)
        tgb = build_reference_recipe_bank(pre, args.bank, args.feat, args.classes, args.dim, device=dev)
        h = pre.inference(feats, adj)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    torch.random.set_rng_state(state)
    kn = tgb.keys_normalized
    index = K.KeyIndex(kn, dedup=dedup)
    out = {"bank_rows": int(kn.shape[0]), "bank_build_s": round(build_s, 2)}
    for B in batches:
        B = min(B, h.shape[0])
        q = h[:B].contiguous()
        for _ in range(3):          # (overflow counts arrive one call late: let the dispatch settle before timing)
            s, i = index.topk(q, args.k)
            torch.cuda.synchronize()
        n = reps or (3 if B > 4096 else 20)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            s, i = index.topk(q, args.k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        inner = index.search_index
        out[f"B{B}"] = {"ms": round(ms, 4), "queries_per_s": round(B / ms * 1e3, 1),
                        "filter": "off (fp32 kernels)" if inner._filter_off or not K.filter_helps(B, inner.keys_normalized.shape[0], args.dim, args.k)
                        else ("bf16 levels" if inner._i8_off or not inner._i8_ok else "int8 levels")}
    rows = torch.linspace(0, B - 1, min(64, B), device=dev).long()
    s32, i32 = K.topk_cosine(q[rows].contiguous(), kn, args.k)
    if not (torch.equal(i[rows], i32) and torch.equal(s[rows], s32)):
        raise SystemExit("bench.py: retrieval on the reference-shaped bank differs from the fp32 kernel over every row")
    top_is_copy = float((i[:, args.k - 1] - i[:, 0] == args.k - 1).float().mean()) if args.k > 1 else 0.0
    if index.duplicate_stats is not None:
        out.update(unique_rows=index.duplicate_stats[1], largest_group=index.duplicate_stats[2])
    out.update(collapsed=bool(index._collapsed), overflowed_queries=index.overflowed_queries,
               verified_rows_vs_fp32_kernel=int(rows.numel()),
               queries_answered_by_k_consecutive_rows=round(top_is_copy, 4))
    return out


def host_cpu_info():
    """What the CPU baseline ran on: cores this process may use (affinity mask, capped by a cgroup CPU quota when one is
    set -- os.cpu_count() reports the machine's, not the container's), CPU model, torch's BLAS."""
    import re

    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    cores = max(1, min(affinity, int(quota + 0.5)) if quota else affinity)
    model = sockets = None
    try:
        info = open("/proc/cpuinfo").read()
        m = re.search(r"model name\s*:\s*(.+)", info)
        model = m.group(1).strip() if m else None
        sockets = len(set(re.findall(r"physical id\s*:\s*(\d+)", info))) or None
    except OSError:
        pass
    cfg = torch.__config__.show()
    blas = re.search(r"BLAS_INFO=(\w+)", cfg)
    par = torch.__config__.parallel_info()
    mkl = re.search(r"Math Kernel Library Version ([^\n]+?) for", par)
    return {"cores_used": cores, "affinity_cores": affinity, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(),
            "cpu_model": model, "sockets": sockets,
            "torch_blas": (blas.group(1) if blas else "?") + (f" ({mkl.group(1).strip()})" if mkl else "")}


def cpu_baseline(args, model, feats, adj):
    """The reference's op chain on the host cores (oracle/ref_torch.py), BASELINE.md section 3: GNN part on the whole
    graph (sparse CSR: the reference's dense adjacency would be 40 GB); retrieval exactly as the reference computes it
    (bank re-normalised on every call, the B x N slab materialised, torch.topk, gathers) on ONE slab of 1024 queries --
    1 warm-up + 3 timed repetitions, median -- extrapolated to the full forward; the 'fair' row with the bank normalised
    once; and the per-phase split of one slab (normalise / GEMM / top-k / gather) with the GEMM's GFLOP/s, so that the
    number can be judged against the cores it ran on.  Threads = the cores this process may USE (host_cpu_info: affinity
    mask and cgroup quota; os.cpu_count() is the machine's)."""
    from oracle import ref_torch

    host = host_cpu_info()
    cores = host["cores_used"]
    torch.set_num_threads(cores)
    n = feats.shape[0]
    slab = min(args.cpu_slab, n)
    conv = model.pretrain_model.gcn.convs[0]
    p = {"W": conv.fc.weight.detach().cpu(), "bias": conv.bias.detach().cpu(), "alpha": conv.act.weight.detach().cpu()}
    tgb = model.toy_graph_base
    keys, vals, labs = tgb.resource_keys.cpu(), tgb.resource_values.cpu(), tgb.resource_labels.cpu()
    adj_cpu = torch.sparse_csr_tensor(adj.rowptr.cpu(), adj.col.cpu().long(), adj.val.cpu(), (n, n))
    X = feats.cpu()

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts), ts

    with torch.no_grad():
        h = ref_torch.gcn_layer(X, adj_cpu, p["W"], p["bias"], p["alpha"])
        t_gnn, ts_gnn = timed(lambda: ref_torch.propagate(adj_cpu, ref_torch.gcn_layer(X, adj_cpu, p["W"], p["bias"], p["alpha"]),
                                                          model.query_graph_hop), 1, 3)
        q = h[:slab].contiguous()
        t_ref, ts_ref = timed(lambda: ref_torch.retrieve(q, keys, vals, labs, args.k, slab=slab), 1, 3)
        kn = torch.nn.functional.normalize(keys, p=2, dim=-1)

        def fair():
            S = torch.matmul(torch.nn.functional.normalize(q, p=2, dim=-1), kn.t())
            _, idx = torch.topk(S, args.k, largest=True, sorted=True)
            return vals[idx].sum(dim=1), labs[idx].mean(dim=1)
        t_fair, ts_fair = timed(fair, 1, 3)
        # one slab, phase by phase (the same ops as ref_torch.retrieve, timed apart; one warm repetition each)
        phases = {}
        tp = time.perf_counter()
        kn2 = torch.nn.functional.normalize(keys, p=2, dim=-1)
        qn = torch.nn.functional.normalize(q, p=2, dim=-1)
        phases["normalize_bank_and_queries_s"] = time.perf_counter() - tp
        tp = time.perf_counter()
        S = torch.matmul(qn, kn2.t())
        phases["score_gemm_s"] = time.perf_counter() - tp
        tp = time.perf_counter()
        _, idx = torch.topk(S, args.k, largest=True, sorted=True)
        phases["topk_s"] = time.perf_counter() - tp
        tp = time.perf_counter()
        vals[idx].sum(dim=1), labs[idx].mean(dim=1)
        phases["gather_reduce_s"] = time.perf_counter() - tp
        gemm_gflops = 2.0 * slab * keys.shape[0] * keys.shape[1] / phases["score_gemm_s"] / 1e9
        del S, kn2
    est_full = t_gnn + t_ref * (n / slab)
    est_fair = t_gnn + t_fair * (n / slab)
    return {"value": round(n / est_full, 2), "unit": "queries/s", "cores": cores, "kind": "port",
            "cpu_model": host["cpu_model"], "affinity_cores": host["affinity_cores"], "cgroup_cpu_quota": host["cgroup_cpu_quota"],
            "os_cpu_count": host["os_cpu_count"], "sockets": host["sockets"], "torch_blas": host["torch_blas"],
            "torch_threads": torch.get_num_threads(),
            "sample": f"GNN encode+{model.query_graph_hop}-hop on all {n} nodes ({t_gnn:.2f}s, torch sparse CSR, median of 3) + retrieval of "
                      f"one slab of {slab} of the {n} queries vs the full {keys.shape[0]}x{keys.shape[1]} bank, 1 warm-up + 3 "
                      f"repetitions, median {t_ref:.2f}s (bank re-normalised per call as the reference does), extrapolated "
                      f"to {n} queries",
            "retrieval_only_queries_per_s": round(slab / t_ref, 2),
            "retrieval_rep_seconds": [round(t, 3) for t in ts_ref],
            "phases_one_slab": {**{k_: round(v, 4) for k_, v in phases.items()}, "score_gemm_GFLOPs": round(gemm_gflops, 1),
                                "gnn_all_nodes_s": round(t_gnn, 3), "gnn_rep_seconds": [round(t, 3) for t in ts_gnn],
                                "gnn_nodes_per_s": round(n / t_gnn, 1)},
            "fair": {"value": round(n / est_fair, 2), "retrieval_only_queries_per_s": round(slab / t_fair, 2),
                     "rep_seconds": [round(t, 3) for t in ts_fair],
                     "note": "bank normalised once (ref_torch.retrieve(renormalize_bank=False) arithmetic), 1 warm-up + 3 "
                             "repetitions, median: what the reference would do without its redundant per-call "
                             "F.normalize(resource_keys)"}}


def verify_retrieval(model, feats, adj, args, world, with_oracle):
    """Outside the timed region: the retrieval the step just timed (product dispatch: bf16-filtered at this shape), ALL
    queries again, and a fixed sample of 64 of its rows against (a) the exact fp32 kernel -- identical indices and score
    bits required -- and (b), 4 rows, the CPU oracle (oracle/cref.py; N = 1 only: it needs the whole bank on the host).
    Key-sharded runs compare the merged global lists with the merge of per-shard fp32 lists.  Raises on any difference."""
    from ragraph_amd import kernels as K
    from ragraph_amd.sharded import all_gather_into

    k = args.k
    tgb = model.toy_graph_base
    with torch.no_grad():
        h = model.pretrain_model.inference(feats, adj)
        n = h.shape[0]
        R = min(VERIFY_ROWS, n)
        rows = torch.linspace(0, n - 1, R, device=h.device).long()
        s_all, i_all = tgb.topk(h, k)                                  # the timed path, every query
        hs = h[rows].contiguous()
        sharded = hasattr(tgb, "idx_base") and getattr(tgb, "collective", False)
        if sharded:
            s32, i32 = K.topk_cosine(hs, tgb.keys_normalized, k, idx_base=tgb.idx_base)
            gw = tgb.world   # (the ranks that hold the shards: the whole job, or this rank's key group in the hybrid layout)
            gs = torch.empty((gw * R, k), dtype=s32.dtype, device=s32.device)
            gi = torch.empty((gw * R, k), dtype=i32.dtype, device=i32.device)
            all_gather_into(gs, s32.contiguous(), tgb.group)
            all_gather_into(gi, i32.contiguous(), tgb.group)
            s32, i32 = K.topk_merge(gs.view(gw, R, k), gi.view(gw, R, k))
        else:
            s32, i32 = K.topk_cosine(hs, tgb.keys_normalized, k)
    ok_i, ok_s = torch.equal(i_all[rows], i32), torch.equal(s_all[rows], s32)
    if not (ok_i and ok_s):
        raise SystemExit(f"bench.py: the timed retrieval path differs from the exact fp32 kernel on the verification sample "
                         f"(indices equal: {ok_i}, scores equal: {ok_s})")
    ix = getattr(tgb, "_index", None)
    ix = getattr(ix, "search_index", ix)
    rec = {"rows": R, "of_queries": n, "vs": ["fp32_kernel"], "identical": True,
           "prior_in_force": getattr(ix, "last_prior", None) if ix is not None else None,
           "what": "top-k indices and score bits of the timed (MFMA-filtered) retrieval, re-run on all queries of the LAST "
                   f"timed batch after the timed region (same dispatch state: the learnt first bound stays in force); {R} "
                   "evenly spaced rows against ragraph_topk_cosine_f32"}
    if with_oracle and not sharded:
        from oracle import cref

        orows = rows[:: max(R // 4, 1)][:4]    # (4 rows spread over the batch: 1 M x 256 keys each on the host)
        os_, oi = cref.topk_cosine(h[orows].cpu().numpy(), tgb.keys_normalized.cpu().numpy(), k)
        if not ((i_all[orows].cpu().numpy() == oi).all() and (s_all[orows].cpu().numpy() == os_).all()):
            raise SystemExit("bench.py: the timed retrieval path differs from the CPU oracle on the verification sample")
        rec["vs"].append("oracle")
        rec["oracle_rows"] = int(orows.numel())
    return rec


def _group_prior_report(model):
    """A row-sharded bank's speculative first bound (ragraph_amd.sharded.GroupPrior): calls of the group so far, how many ran
    without bound pass / phase 0, how many were repeated because a row missed the prior."""
    tgb = getattr(model, "toy_graph_base", None)
    pr = getattr(tgb, "prior", None)
    if pr is None or not hasattr(pr, "calls"):
        return None
    return {"group_calls": pr.calls, "speculative_calls": pr.used, "calls_repeated_after_a_miss": getattr(tgb, "reruns", 0),
            "phase0_exchanges": getattr(tgb, "exchange_count", {}).get(0, 0)}


def timed_steps(step, steps, world, dev, on_step=None):
    """Exactly `steps` forwards between barrier + synchronize on both sides; MAX over ranks."""
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step()
        if on_step:
            on_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        from ragraph_amd.sharded import all_reduce   # (stages device tensors through the host under --backend gloo)

        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torchrun job (nothing in this
        # process has touched the GPU yet -- never re-exec after it has) and exit with its code.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"),
               os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in ragraph_amd)")
    if args.backend == "gloo":  # (functional check: the ranks may share a device)
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("RAGRAPH_FORCE_DIST") == "1"  # 1-rank RCCL group: exercises the sharded path on one GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=_timeout(args))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=_timeout(args))

    from ragraph_amd import kernels as K

    beat = lambda phase: None   # noqa: E731
    hb = ranks_seen = None
    if world > 1 or force_dist:
        from ragraph_amd import sharded as SH

        hb = Heartbeat(rank, world, args.dist_timeout)
        _ACTIVE["hb"] = hb
        beat = hb
        SH.heartbeat = hb
        ranks_seen = preflight(rank, world, dev, args.backend, beat)   # BEFORE the bank is built
    beat("build workload")

    real_filter_helps = K.filter_helps
    if args.exact_fp32:
        K.filter_helps = lambda *a, **kw: False
    topk_timer = EventTimer(K, "topk_cosine")             # fp32 kernel (the whole retrieval with --exact-fp32)
    filt_timer = EventTimer(K, "topk_cosine_filtered")    # bound pass + bf16 filter + rescoring
    model, feats, adj, n_local = build_workload(args, dev, rank, world, args.shard, force_dist)
    L = K.N.lib()
    prof = L.ragraph_filter_profile_create()   # caller-owned; attached to this thread for the timed region
    L.ragraph_filter_profile_attach(prof)
    filter_ms = []
    level_ms = []   # per step: [(slot, ms, int8?, keys)] of the call's filter launches (slot 3 = the bound pass)

    # Distinct batches: the speculative first bound of the filtered call is learnt from the statistics of EARLIER calls, so a
    # repeated batch could never miss it.  `--batches` feature tensors (seeds 4321, 4322, ...; same distribution) are cycled
    # through warm-up and timed steps; with more batches than warm-up steps the timed region meets batches no call has seen.
    nb = max(1, args.batches)
    feats_all = [feats] + [torch.randn(args.nodes, args.feat, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + b_))
                           for b_ in range(1, nb)]
    step_no = [0]
    seen_in_warmup = set()
    step_log = []   # per timed step: (batch, prior in force or None, clone of the call's statistics words or None)

    def step():
        f = feats_all[step_no[0] % nb]
        step_no[0] += 1
        with torch.no_grad():
            return model(f, adj)

    def _index_of(tgb_):
        ix = getattr(tgb_, "_index", None)
        return getattr(ix, "search_index", ix) if ix is not None else None

    def grab_filter_ms():
        ix = _index_of(model.toy_graph_base)
        st_ = filt_timer.last[3].clone() if isinstance(filt_timer.last, tuple) and len(filt_timer.last) == 4 else None
        step_log.append(((step_no[0] - 1) % nb, getattr(ix, "last_prior", None) if ix is not None else None, st_))
        ms = L.ragraph_filter_profile_last_ms(prof)  # waits for this step's filter launches (they are the step's tail)
        if ms > 0:
            filter_ms.append(ms)
            a_ms, a_i8, a_keys = (ctypes.c_float * 4)(), (ctypes.c_int * 4)(), (ctypes.c_int64 * 4)()
            L.ragraph_filter_profile_levels(prof, a_ms, a_i8, a_keys)
            level_ms.append([(s_, float(a_ms[s_]), int(a_i8[s_]), int(a_keys[s_])) for s_ in range(4) if a_ms[s_] > 0])

    if os.environ.get("RAGRAPH_BENCH_HANG_RANK") == str(rank):   # (test hook: this rank stops taking part -- the watchdogs must end the job)
        beat("hung on purpose (RAGRAPH_BENCH_HANG_RANK)")
        time.sleep(10 ** 6)
    for w_ in range(args.warmup):
        beat(f"warm-up step {w_}")
        seen_in_warmup.add(step_no[0] % nb)
        step()
        torch.cuda.synchronize()   # (untimed: the dispatch settles -- overflow counts and the speculative first bound's
                                   # statistics arrive behind an event and are read at the NEXT call)
    topk_timer.enabled = filt_timer.enabled = True
    beat("timed steps")
    elapsed, out = timed_steps(step, args.steps, world, dev, grab_filter_ms)
    topk_timer.enabled = filt_timer.enabled = False
    collectives = None
    if world > 1 or force_dist:
        # what its collectives cost: one more step OUTSIDE the timed region with every collective of ragraph_amd.sharded
        # bracketed by events on the stream it is ordered on (what the process group is: `ranks_seen`, the pre-flight above)
        from ragraph_amd import sharded as SH

        beat("collective timing step")
        SH.collective_times.reset()
        SH.collective_times.enabled = True
        step()
        collectives = SH.collective_times.report()
        SH.collective_times.enabled = False
    L.ragraph_filter_profile_attach(None)
    L.ragraph_filter_profile_destroy(prof)
    # candidates per query of the last timed call's levels (sampled by the call itself: every 64th query; the 16 ints it
    # leaves at the end of its workspace, include/ragraph_hip.h)
    last = filt_timer.last if isinstance(filt_timer.last, tuple) and len(filt_timer.last) == 4 else None
    cand_levels = K.filter_stats_levels(last[3].cpu().tolist()) if last is not None else []
    assert torch.isfinite(out).all()
    verified = None
    if args.emulate_rank_of <= 1 and not args.exact_fp32 and not args.no_extras:
        # (every rank: the sharded check has collectives; profiling runs -- --no-extras -- keep their kernel lists clean)
        verified = verify_retrieval(model, feats_all[(step_no[0] - 1) % nb], adj, args, world,
                                    with_oracle=(rank == 0 and world == 1 and not args.no_cpu_baseline))

    n = args.nodes
    G = max(world, args.emulate_rank_of, 1)
    S_key = max(1, min(args.key_shards, G)) if args.shard == "hybrid" else (G if args.shard == "keys" else 1)
    shard_div = G if args.shard == "queries" else (G // S_key if args.shard == "hybrid" else 1)
    n_q_local = -(-n // shard_div)  # queries this rank scores against its n_local keys
    traffic = None  # HBM-side GB per launch from the committed PMC run of this exact shape (cannot be sampled in-process)
    try:
        prof = json.load(open(TRAFFIC_JSON))
        filtered_shape = not args.exact_fp32 and real_filter_helps(n_q_local, n_local, args.dim, args.k)
        key = (f"topk_filter_kernel B={n_q_local} N={n_local} D={args.dim} k={args.k}" if filtered_shape else
               f"topk_stream_kernel<{args.dim}> B={n_q_local} N={n_local} D={args.dim} k={args.k}")
        if key in prof:
            traffic = prof[key]["hbm_side_GB"]
    except (OSError, ValueError):
        pass
    ms_step = elapsed / args.steps * 1e3
    flops = 2.0 * n_q_local * n_local * args.dim
    filtered = len(filt_timer.events) > 0
    traffic_unit = ("GB per launch, HBM side = (2*FETCH_SIZE + WRITE_SIZE): the COMMITTED rocprofv3 --pmc record for this "
                    "shape (separate passes, profiles/" + os.path.basename(TRAFFIC_JSON) + "), looked up by shape key -- "
                    "NOT sampled in this run (counters cannot be read in-process)")
    if filtered:
        # Dominant kernel = the filter kernel (ragraph::topk_filter_kernel), timed per launch by events the library records
        # around its launches on the launch stream.  A call launches it once per level -- the first on the bf16 copy
        # (v_mfma_f32_16x16x32_bf16), the later ones on the int8 copy (v_mfma_i32_16x16x64_i8, twice the rate per clock) --
        # plus the bound pass over a prefix; `roofline` is the launch that takes longest, priced against the peak of ITS
        # matrix instruction; `levels` lists them all, `whole_call` the algorithmic 2 B N D over their sum.
        kernel_ms = sum(filter_ms) / max(len(filter_ms), 1)
        call_ms = filt_timer.mean_ms()
        per = {}
        for step_levels in level_ms:
            for slot, ms, i8, keys in step_levels:
                per.setdefault(slot, {"ms": [], "i8": i8, "keys": keys})["ms"].append(ms)
        levels = []
        for slot in sorted(per):
            e = per[slot]
            ms = sum(e["ms"]) / len(e["ms"])
            ops = 2.0 * n_q_local * e["keys"] * args.dim
            peak = INT8_MFMA_PEAK_TOPS if e["i8"] else BF16_MFMA_PEAK_TFLOPS
            levels.append({"launch": "bound pass (prefix)" if slot == 3 else f"level {slot + 1}", "dtype": "int8" if e["i8"] else "bf16",
                           "keys": e["keys"], "ms": round(ms, 3), "achieved": round(ops / (ms * 1e-3) / 1e12, 1), "peak": peak,
                           "frac": round(ops / (ms * 1e-3) / 1e12 / peak, 4)})
            if slot < 3 and slot < len(cand_levels) and cand_levels[slot][2] is not None:
                levels[-1]["candidates_per_query"] = round(cand_levels[slot][2], 1)
        dom = max((lv for lv in levels if not lv["launch"].startswith("bound")), key=lambda lv: lv["ms"], default=None)
        whole = flops / (kernel_ms * 1e-3) / 1e12
        if dom is None:
            dom = {"dtype": "bf16", "ms": kernel_ms, "achieved": round(whole, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                   "frac": round(whole / BF16_MFMA_PEAK_TFLOPS, 4), "launch": "all", "keys": n_local}
        insn = "v_mfma_i32_16x16x64_i8 on the int8 copy" if dom["dtype"] == "int8" else "v_mfma_f32_16x16x32_bf16 on the bf16 copy"
        roofline = {
            "kernel": f"ragraph::topk_filter_kernel, {dom['launch']} of the exact top-k's filter ({insn})",
            "bound": "mfma", "achieved": dom["achieved"], "peak": dom["peak"], "unit": "TFLOP/s",
            "frac": dom["frac"], "traffic": traffic, "traffic_unit": traffic_unit,
            "launch_ms": dom["ms"], "dtype": dom["dtype"],
            "levels": levels,
            "whole_call": {"filter_ms": round(kernel_ms, 3), "algorithmic_TFLOPs": round(whole, 1),
                           "vs_bf16_peak": round(whole / BF16_MFMA_PEAK_TFLOPS, 4),
                           "what": "2*B*N*D of the whole score matrix / the summed duration of the call's filter launches "
                                   "(bound pass + the levels)"},
            "note": "achieved = 2*B*keys*D operations of the launch that takes longest (for the int8 levels: integer "
                    "multiply-adds counted as 2, against the int8 dense peak = twice the bf16 one) / its mean duration from "
                    "events recorded around it inside the library on the launch stream.  The whole exact retrieval call "
                    f"(every filter launch + exact fp32 rescoring of the survivors) takes {call_ms:.2f} ms; `exact_fp32` "
                    "below is the same step on the fp32 MFMA kernel alone.  The bare inner loops sustain 1.68 PFLOP/s (bf16) "
                    "and 3.29 Pop/s (int8) on random operands (tools/microbench/mfma_i8_bench.hip): the clock the chip holds "
                    "under real data bounds both well below the nominal peaks",
            "retrieval_call_ms": round(call_ms, 3),
        }
    else:
        topk_ms = topk_timer.mean_ms()
        achieved = flops / (topk_ms * 1e-3) / 1e12
        roofline = {
            "kernel": "ragraph::topk_stream_kernel<256, 4> (fused cosine+top-k, v_mfma_f32_32x32x2_f32, LDS-DMA key ring)",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": traffic_unit,
            "launch_ms": round(topk_ms, 3),
            "note": "algorithmic flops 2*B*N*D per launch / mean launch time from events on the launch stream "
                    "(includes the <0.1 % query-normalise and select kernels of the same ABI call)",
        }
    if args.shard == "keys":
        par = (f"key bank row-sharded x{G} (values replicated): per-query bounds pooled at every phase of the filtered call "
               f"(all_gather of each rank's best 2*ceil(k/G) lower bounds + k-th of the union), one all_to_all of the "
               f"per-shard lists to the rank that owns the rows, query-sharded tail, one all_gather of the [n, C] outputs "
               f"({args.backend})")
    elif args.shard == "hybrid":
        par = (f"{G // S_key} query groups x {S_key} key shards (rank = q * {S_key} + s): a rank scores its group's slice of the "
               f"queries against its key shard, bound exchanges and the all_to_all of the lists stay inside the key group, two "
               f"all_gathers of [., C] outputs (key group, query axis) ({args.backend})")
    else:
        par = (f"query batch split x{G}, bank replicated on every GPU (1 GB of 288 GB): no data-path collective, one RCCL "
               f"all_gather of the [n, C] outputs per step")
    layout = {"keys": f"key-sharded x{G}", "queries": f"query-sharded x{G}",
              "hybrid": f"hybrid {G // S_key}x{S_key} (query groups x key shards)"}[args.shard] if G > 1 else "single GPU"
    result = {
        "metric": "retrieved-queries/sec (RAGraph_node forward: GCN encode + cosine/top-k retrieval + 3-hop propagate + decode)",
        "value": round(n / (elapsed / args.steps), 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if args.exact_fp32 else "f32 (exact results; candidates pre-filtered on bf16 / int8 MFMA with proven bounds)",
        "data": "synthetic",
        "config": {"workload": f"RAGraph_node forward, synthetic {n}-node graph (F={args.feat}, mean degree ~10), "
                               f"{args.bank}-key x {args.dim}-d bank, k={args.k}, C={args.classes} "
                               f"(BASELINE.json configs[1])",
                   "bank_rows_per_gpu": n_local, "queries_per_gpu": n_q_local, "layout": layout,
                   "parallelism": "single GPU" if G == 1 else par},
        "roofline": roofline,
    }
    # the speculative first bound over the timed steps: which prior each step filtered under and how many of its queries the
    # verify launch sent to the exact scan because their k-th best fell below it (statistics word 17; word 20 = every query
    # answered by a scan, overflowed lists included).  (Read here, after the timed region: the words were cloned on the stream.)
    timed_batches = sorted({b_ for b_, _, _ in step_log})
    spec = {"distinct_batches": nb, "timed_batches": timed_batches,
            "timed_batches_unseen_in_warmup": sorted(set(timed_batches) - seen_in_warmup)}
    if step_log and any(st_ is not None for _, _, st_ in step_log):
        words = [st_.cpu().tolist() if st_ is not None else None for _, _, st_ in step_log]
        spec["prior_per_step"] = [None if pr_ is None else round(float(pr_), 5) for _, pr_, _ in step_log]
        spec["speculative_steps"] = sum(1 for w_ in words if w_ is not None and w_[16] != 0)
        spec["misses_per_step"] = [None if w_ is None else int(w_[17]) for w_ in words]
        spec["scanned_queries_per_step"] = [None if w_ is None else int(w_[20]) for w_ in words]
        if filt_timer.events:
            calls = [a.elapsed_time(b) for a, b in filt_timer.events]
            spec["retrieval_call_ms_min_max"] = [round(min(calls), 3), round(max(calls), 3)]
    result["distinct_batches"] = nb
    result["first_bound"] = spec
    gp = _group_prior_report(model)
    if gp is not None:
        result["first_bound"]["group"] = gp
    if ranks_seen is not None:
        result["ranks_seen"] = ranks_seen
        result["collectives_per_step"] = collectives
    if args.emulate_rank_of > 1:
        result["emulated"] = (f"rank 0 of a {args.emulate_rank_of}-GPU job ({args.shard}-sharded), collectives replaced by "
                              f"their local part: value is NOT a job throughput")
    extras = not args.no_extras and args.emulate_rank_of <= 1
    if world > 1 and extras:
        # the other layouts, right after (every rank must take part): bank replicated + query batch split, and -- from 4
        # ranks -- query groups x key shards
        del model
        torch.cuda.empty_cache()
        others = [sh for sh in ("keys", "queries", "hybrid") if sh != args.shard and
                  (sh != "hybrid" or (world >= 4 and world % args.key_shards == 0 and 1 < args.key_shards < world))]
        if hb is not None:
            hb.headline = dict(result, verified=verified) if verified is not None else dict(result)
        for sh in others:
            beat(f"layout {sh}: build")
            m2, f2, a2, nl2 = build_workload(args, dev, rank, world, sh, force_dist)

            def step2():
                with torch.no_grad():
                    return m2(f2, a2)
            for w_ in range(max(args.warmup, 1)):
                beat(f"layout {sh}: warm-up step {w_}")
                step2()
            beat(f"layout {sh}: timed steps")
            e2, _ = timed_steps(step2, args.steps, world, dev)
            from ragraph_amd import sharded as SH   # what this layout's collectives cost: one more step, outside its timing
            SH.collective_times.reset()
            SH.collective_times.enabled = True
            step2()
            coll2 = SH.collective_times.report()
            SH.collective_times.enabled = False
            name = {"keys": "key_sharded", "queries": "query_sharded", "hybrid": "hybrid"}[sh]
            result[name] = {"value": round(n / (e2 / args.steps), 1), "unit": "queries/s",
                            "ms_per_step": round(e2 / args.steps * 1e3, 3), "bank_rows_per_gpu": nl2,
                            "layout": getattr(m2, "layout_name", f"{name.replace('_', '-')} x{world}"),
                            "collectives_per_step": coll2, "first_bound": _group_prior_report(m2),
                            "note": "same steps / warm-up as the headline layout"}
            del m2, f2, a2
            torch.cuda.empty_cache()
    if world == 1 and extras:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_blocks as BB

        K.filter_helps = real_filter_helps
        cpu = None
        if rank == 0 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args, model, feats, adj)
        cores = cpu["cores"] if cpu else host_cpu_info()["cores_used"]
        # both halves of BASELINE.json's metric: `value` above = retrieved queries/s of the whole forward; the GNN forward:
        gnn = BB.gnn_fwd_block(model, feats, adj, reps=max(args.steps, 30),
                               cpu_gnn_s=cpu["phases_one_slab"]["gnn_all_nodes_s"] if cpu else None, cpu_cores=cores)
        gnn["structured_graph"] = BB.structured_graph_row(args.feat, args.dim, model.query_graph_hop, dev)
        if nb > 1:   # round 5's figure: ONE batch replayed (the learnt first bound cannot miss by construction)
            def step_same():
                with torch.no_grad():
                    return model(feats, adj)
            for _ in range(2):
                step_same()
            e_rep, _ = timed_steps(step_same, min(args.steps, 10), 1, dev)
            result["repeated_batch"] = {"ms_per_step": round(e_rep / min(args.steps, 10) * 1e3, 3), "steps": min(args.steps, 10),
                                        "value": round(n / (e_rep / min(args.steps, 10)), 1),
                                        "note": "the same step with ONE feature tensor replayed (rounds 1-5 timed this)"}
        result["gnn_fwd"] = gnn
        result["gnn_fwd_nodes_per_s"] = gnn["nodes_per_s"]
        result["retrieval_small_batch"] = small_batch_rates(model.toy_graph_base, args.dim, args.k, dev)
        result["retrieval_reference_bank"] = reference_bank_rates(args, dev, adj, feats)
        result["memory"] = BB.memory_block(model.toy_graph_base)
        if not args.exact_fp32:
            # the same step on the fp32 MFMA kernels alone (the path the bf16 filter replaces bit for bit)
            K.filter_helps = lambda *a, **kw: False
            topk_timer.reset()
            step()
            topk_timer.enabled = True
            e32, _ = timed_steps(step, 2, 1, dev)
            topk_timer.enabled = False
            K.filter_helps = real_filter_helps
            t32 = topk_timer.mean_ms()
            a32 = flops / (t32 * 1e-3) / 1e12
            result["exact_fp32"] = {"value": round(n / (e32 / 2), 1), "unit": "queries/s", "ms_per_step": round(e32 / 2 * 1e3, 3),
                                    "steps": 2,
                                    "roofline": {"kernel": "ragraph::topk_stream_kernel<256, 4> (v_mfma_f32_32x32x2_f32)",
                                                 "bound": "mfma", "achieved": round(a32, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                                                 "unit": "TFLOP/s", "frac": round(a32 / FP32_MFMA_PEAK_TFLOPS, 4),
                                                 "launch_ms": round(t32, 3)}}
        if not args.no_configs:
            # the other configs of BASELINE.json (c1, c3, few-shot, c5-shaped single-GPU leg) and the fine-tuning steps, each
            # with its rate and the roofline that bounds it (tools/bench_blocks.py)
            ft = {"node_528": BB.finetune_node(dev, cores, "528", cpu=cpu is not None),
                  "node_c2": BB.finetune_node(dev, cores, "c2", c2=(model, feats, adj), cpu=cpu is not None)}
            result["configs"], m5 = BB.configs_block(dev)
            ft["edge_c5"] = BB.finetune_edge(dev, cores, m5, cpu=cpu is not None)
            del m5
            result["finetune_step"] = ft
        if cpu is not None:
            result["cpu_baseline"] = cpu
    if verified is not None:
        result["verified"] = verified
    if hb is not None:
        beat("done")
        hb.stop()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


_ACTIVE = {}   # the running job's Heartbeat, for the report below


def _main_reporting():
    """main(), and when it raises in a multi-rank job (a collective that timed out, a rank that died): every rank's last
    heartbeat on stderr before the exception ends this rank with a non-zero code."""
    try:
        main()
    except BaseException as e:   # noqa: B902  (SystemExit included: a failed check also names the phases)
        hb = _ACTIVE.get("hb")
        if hb is not None and not (isinstance(e, SystemExit) and e.code in (0, None)):
            hb.stop()
            print(json.dumps({"bench_failed": f"rank {hb.rank}: {type(e).__name__} in phase '{hb.phase}'", "ranks": hb.report()}),
                  file=sys.stderr, flush=True)
        raise


if __name__ == "__main__":
    _main_reporting()
