"""The blocks bench.py adds to its JSON line at N = 1 (outside the timed region): every number of BASELINE.json's metric
and every other config of it, each timed with events on the launch stream and priced against the roofline that bounds it.

  gnn_fwd        the second half of the metric (GNN-forward nodes/s): encode and encode + k hops, SURVEY section 8(d)'s byte
                 models, the counter bytes of the same kernels (profiles/r5_pmc_traffic.json), the CPU port's rate, and a
                 graph WITH structure (community graph) in its shuffled numbering and after the automatic reordering
  configs        c1 (Cora-shaped node forward), c3 (PROTEINS-style graph batches), the few-shot node forward, c5 (edge
                 flavour: one 4096-query slab and 256 queries against the 4M x 64 bank, and generate() over all nodes)
  finetune_step  forward + loss + backward + Adam of the node flavour (528-node batches / the c2 graph) and the edge
                 flavour's cal_loss step, beside the torch-CPU restatement of the same step
  memory         bytes of every bank image the c2 model holds

Imported by bench.py only (and by tools/bench_configs.py for stand-alone runs); nothing here is product code.
"""
from __future__ import annotations

import json
import os
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6300.0
FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2516.6
INT8_MFMA_PEAK_TOPS = 5033.2
TRAFFIC_JSON = next((p for p in (os.path.join(ROOT, "profiles", f"r{r}_pmc_traffic.json") for r in (6, 5)) if os.path.exists(p)),
                    os.path.join(ROOT, "profiles", "r5_pmc_traffic.json"))


def event_ms(fn, reps, warm=3):
    """Mean ms per call from two events on the current stream around `reps` back-to-back calls."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def capture(fn):
    """fn captured in a HIP graph (two eager runs on a side stream first); returns the replay callable."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def _traffic(key):
    try:
        return json.load(open(TRAFFIC_JSON)).get(key)
    except (OSError, ValueError):
        return None


# ---- GNN forward ------------------------------------------------------------------------------------------------------
def gnn_bytes(n, F, D, nnz, hops, one_launch=False):
    """SURVEY section 8(d), GNN forward.  `ideal`: every feature row read once (ideal reuse) -- the encoder as this
    implementation associates it, (A X) W^T: aggregation over the NARROW features (CSR + 4 n F read + 4 n F written), the
    dense part (4 n F read + the weights + 4 n D written), and per hop CSR + 4 n D read + 4 n D written.  `no_reuse`: the
    third term of every aggregation replaced by 4 nnz width (each neighbour row fetched once per edge).  one_launch: the
    encoder is ragraph_spmm_linear_f32 -- the aggregated table is never written or read back (8 n F bytes fewer)."""
    csr = nnz * 8 + 8 * (n + 1)
    table = 0 if one_launch else 8 * n * F
    enc_agg, dense, hop = csr + 4 * n * F + table, 4 * F * D + 4 * n * D, csr + 8 * n * D
    enc_agg_nr, hop_nr = csr + 4 * nnz * F + table, csr + 4 * nnz * D + 4 * n * D
    return {"encode_ideal": enc_agg + dense, "hop_ideal": hop, "forward_ideal": enc_agg + dense + hops * hop,
            "forward_no_reuse": enc_agg_nr + dense + hops * hop_nr, "hop_no_reuse": hop_nr, "encode_flops": 2.0 * n * F * D + 2.0 * nnz * F,
            "hop_flops": 2.0 * nnz * D}


def _gnn_times(pre, feats, adj, hops, reps):
    from ragraph_amd.ragraph_utils import Propagation

    with torch.no_grad():
        h = pre.inference(feats, adj)
        t_enc = event_ms(lambda: pre.inference(feats, adj), reps)
        t_hops = event_ms(lambda: Propagation.aggregate_k_hop_features(adj, h, hops), reps)
        t_full = event_ms(lambda: Propagation.aggregate_k_hop_features(adj, pre.inference(feats, adj), hops), reps)
    return t_enc, t_hops, t_full


def gnn_fwd_block(model, feats, adj, reps=30, cpu_gnn_s=None, cpu_cores=None):
    """GNN-forward nodes/s of the c2 step's graph part (encode = 1 GCN layer; forward = encode + query_graph_hop hops), with
    the roofline that bounds it (HBM: the kernels gather rows; the dense part is 6.6 GFLOP, 42 us at the fp32 MFMA peak)."""
    n, F = feats.shape
    D, hops, nnz = model.emb_size, model.query_graph_hop, adj.nnz
    t_enc, t_hops, t_full = _gnn_times(model.pretrain_model, feats, adj, hops, reps)
    from ragraph_amd import kernels as K
    from ragraph_amd.layers.gcn import aggregate_first

    one_launch = aggregate_first(F, D) and K.spmm_linear_helps(n, F, D)
    b = gnn_bytes(n, F, D, nnz, hops, one_launch)
    rec = {"nodes": n, "nnz": nnz, "feat": F, "dim": D, "hops": hops, "encoder_one_launch": bool(one_launch),
           "ms": round(t_full, 4), "nodes_per_s": round(n / t_full * 1e3, 1),
           "encode_ms": round(t_enc, 4), "encode_nodes_per_s": round(n / t_enc * 1e3, 1),
           "hop_ms": round(t_hops / max(hops, 1), 4),
           "bytes_ideal": b["forward_ideal"], "bytes_no_reuse": b["forward_no_reuse"],
           "GBps": round(b["forward_ideal"] / t_full / 1e6, 1),
           "frac": round(b["forward_ideal"] / t_full / 1e6 / HBM_PEAK_GBS, 4),
           "frac_of_achievable": round(b["forward_ideal"] / t_full / 1e6 / HBM_ACHIEVABLE_GBS, 4),
           "GBps_no_reuse": round(b["forward_no_reuse"] / t_full / 1e6, 1),
           "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "hop": {"bytes_ideal": b["hop_ideal"], "GBps": round(b["hop_ideal"] / (t_hops / hops) / 1e6, 1),
                   "frac": round(b["hop_ideal"] / (t_hops / hops) / 1e6 / HBM_PEAK_GBS, 4)},
           "encode": {"bytes_ideal": b["encode_ideal"], "GBps": round(b["encode_ideal"] / t_enc / 1e6, 1),
                      "frac": round(b["encode_ideal"] / t_enc / 1e6 / HBM_PEAK_GBS, 4),
                      "dense_TFLOPs": round(2.0 * n * F * D / (t_enc * 1e-3) / 1e12, 2)},
           "what": "bytes_ideal = SURVEY 8(d)'s ideal-reuse model (CSR + every feature row read once + the result written, per "
                   "aggregation; + the dense part's operands) of encode + hops; frac = bytes_ideal / ms / 8 TB/s.  bytes_counter = "
                   "HBM-side bytes (2 FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc in separate passes) of the same kernels, from "
                   f"profiles/{os.path.basename(TRAFFIC_JSON)}: above bytes_ideal by the L2 misses of the row gathers"}
    tr = _traffic(f"gnn_forward n={n} F={F} D={D} hops={hops}")
    rec["bytes_counter"] = None if tr is None else tr.get("hbm_side_bytes")
    if tr is not None:
        rec["bytes_counter_over_ideal"] = round(tr["hbm_side_bytes"] / b["forward_ideal"], 2)
        rec["counter_kernels"] = tr.get("per_kernel")
    if cpu_gnn_s:
        rec["cpu_nodes_per_s"] = round(n / cpu_gnn_s, 1)
        rec["cpu_cores"] = cpu_cores
        rec["cpu_what"] = "oracle/ref_torch.py gcn_layer + propagate on torch sparse CSR, the host cores of cpu_baseline"
    return rec


def locality_probe(g, sample=4096):
    """Cheap device-side probe of whether a reordering can pay: the share of a sample of edges whose two end points lie
    within 2048 rows of each other in the CURRENT numbering (a 4-MiB L2 holds 4096 rows of 256 floats) and the clustering of
    the sampled rows' neighbourhoods (share of a node's neighbour pairs that are neighbours themselves, estimated through
    common-neighbour counts on the sample).  A graph with communities hidden by its numbering shows few near edges but many
    shared neighbours; an Erdos-Renyi graph shows neither.  Two reductions, one read-back."""
    n = g.n
    deg = (g.rowptr[1:] - g.rowptr[:-1])
    rows = torch.repeat_interleave(torch.arange(n, device=g.device), deg)
    pick = torch.randint(0, g.nnz, (min(sample, g.nnz),), device=g.device, generator=torch.Generator(device=g.device).manual_seed(7))
    r, c = rows[pick], g.col[pick].long()
    near = ((r - c).abs() <= 2048).float().mean()
    # shared neighbours of the two ends of a sampled edge (sorted columns: intersect through searchsorted on keys row*n+col)
    keys = rows * n + g.col.long()
    shared = torch.zeros(pick.numel(), device=g.device)
    cap = 16   # neighbours of r looked up in c's row (rows longer than this: the first `cap`)
    for j in range(cap):
        e = g.rowptr[r] + j
        ok = e < g.rowptr[r + 1]
        nb = g.col[e.clamp_max(g.nnz - 1)].long()
        probe = c * n + nb
        pos = torch.searchsorted(keys, probe).clamp_max(g.nnz - 1)
        shared += (ok & (keys[pos] == probe) & (nb != r) & (nb != c)).float()
    mean_deg = deg.float().mean()
    out = torch.stack([near, shared.mean(), mean_deg]).tolist()
    return {"edges_within_2048_rows": round(out[0], 4), "shared_neighbours_per_edge": round(out[1], 3), "mean_degree": round(out[2], 2)}


def reorder_pays(probe) -> bool:
    """The rule the probe feeds: structure is there (the ends of an edge share neighbours: >= 0.5 per edge; an Erdos-Renyi
    graph of mean degree 10 over 1e5 nodes shares 0.001) and the numbering does not show it yet (< half of the edges near)."""
    return probe["shared_neighbours_per_edge"] >= 0.5 and probe["edges_within_2048_rows"] < 0.5


def structured_graph_row(feat, dim, hops, dev, n=100_000, reps=20):
    """GNN forward on a graph WITH structure (512-node communities, 90 % of the edges inside, node ids shuffled as real data
    arrive): in the given numbering, and after CSRGraph.locality_order() applied automatically when locality_probe says it
    pays.  The reordered forward permutes the features once, runs the same kernels, and un-permutes the result; its rows
    equal the natural-order forward's up to fp32 summation order (checked here, 1e-5)."""
    from ragraph_amd.data import synthetic_community_graph
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.ragraph_utils import Propagation

    state = torch.random.get_rng_state()
    torch.manual_seed(2)
    pre = PrePrompt(feat, dim, "prelu", 1, 0.3).to(dev)
    torch.random.set_rng_state(state)
    ei, _ = synthetic_community_graph(n, 10, 512, 0.9, device=dev)
    g = CSRGraph.from_edge_index_sym_normalized(ei, n)
    X = torch.randn(n, feat, device=dev, generator=torch.Generator(device=dev).manual_seed(99))
    probe = locality_probe(g)
    t_enc, t_hops, t_full = _gnn_times(pre, X, g, hops, reps)
    from ragraph_amd import kernels as K
    from ragraph_amd.layers.gcn import aggregate_first

    b = gnn_bytes(n, feat, dim, g.nnz, hops, aggregate_first(feat, dim) and K.spmm_linear_helps(n, feat, dim))
    rec = {"graph": f"{n} nodes, 512-node communities, 90 % intra-community edges, shuffled ids, nnz {g.nnz}",
           "probe": probe, "reorder": reorder_pays(probe),
           "given_order": {"ms": round(t_full, 4), "hop_ms": round(t_hops / hops, 4), "nodes_per_s": round(n / t_full * 1e3, 1),
                           "frac": round(b["forward_ideal"] / t_full / 1e6 / HBM_PEAK_GBS, 4)}}
    if rec["reorder"]:
        t0 = time.perf_counter()
        order = g.locality_order()
        g2 = g.permuted(order)
        _ = g2.row_normalized_values()
        torch.cuda.synchronize()
        rec["reorder_s"] = round(time.perf_counter() - t0, 3)
        X2 = X[order].contiguous()
        t_enc2, t_hops2, t_full2 = _gnn_times(pre, X2, g2, hops, reps)
        with torch.no_grad():
            a = Propagation.aggregate_k_hop_features(g, pre.inference(X, g), hops)
            c = Propagation.aggregate_k_hop_features(g2, pre.inference(X2, g2), hops)
        rec["max_abs_diff_vs_given_order"] = float((c - a[order]).abs().max())
        rec["reordered"] = {"ms": round(t_full2, 4), "hop_ms": round(t_hops2 / hops, 4), "nodes_per_s": round(n / t_full2 * 1e3, 1),
                            "frac": round(b["forward_ideal"] / t_full2 / 1e6 / HBM_PEAK_GBS, 4),
                            "hop_frac": round(b["hop_ideal"] / (t_hops2 / hops) / 1e6 / HBM_PEAK_GBS, 4)}
    return rec


# ---- the other configs ------------------------------------------------------------------------------------------------
def _bank(model, N, D, C, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    model.toy_graph_base.add_resources(torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=-1),
                                       torch.randn(N, D, device=dev, generator=g),
                                       torch.nn.functional.one_hot(torch.randint(0, C, (N,), device=dev, generator=g), C).float())
    _ = model.toy_graph_base.keys_normalized


def _small_forward_roofline(ms, bytes_, flops):
    """A forward of a few hundred to a few thousand nodes: bytes every operand once + flops of its products, each against its
    peak; both fractions are small -- such a forward is a chain of launches, bound by launch latency (stated as such)."""
    t = ms * 1e-3
    return {"bytes_once": int(bytes_), "GBps": round(bytes_ / t / 1e9, 1), "frac_hbm": round(bytes_ / t / 1e9 / HBM_PEAK_GBS, 4),
            "TFLOPs_fp32": round(flops / t / 1e12, 2), "frac_fp32_mfma": round(flops / t / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "bound": "launch latency (a chain of dependent launches; neither roof is near)"}


def config_c1(dev, reps=50):
    """c1: RAGraph_node on a Cora-shaped graph (2708 nodes, F = 1433 bag-of-words), 10k x 128 bank, k = 5."""
    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph

    n, F, D, C, N, k = 2708, 1433, 128, 7, 10_000, 5
    adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 4, seed=7, device=dev), n)
    X = (torch.rand(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(70)) < 0.0127).float()
    X = X / X.sum(1, keepdim=True).clamp_min(1)
    m = RAGraph(PrePrompt(F, D, "prelu", 1, 0.3).to(dev), None, F, C, D, device=dev).eval()
    m.toy_graph_base.retrieve_num = k
    _bank(m, N, D, C, dev, 71)
    _ = adj.row_normalized_values()
    with torch.no_grad():
        f = lambda: m(X, adj)
        te, tg = event_ms(f, reps), event_ms(capture(f), reps)
    nnz_x = int(torch.count_nonzero(X))
    bytes_ = nnz_x * 8 + F * D * 4 + 4 * (adj.nnz * 8 + 8 * n * D) + N * D * 4 + n * k * 4 * (D + C) + 2 * D * D * 4
    flops = 2.0 * nnz_x * D + 4 * 2.0 * adj.nnz * D + 2.0 * n * N * D + 2.0 * n * D * D
    return {"config": f"c1 RAGraph_node, Cora-shaped {n} nodes (F={F}), {N} x {D} bank, k={k}", "eager_ms": round(te, 4),
            "ms": round(tg, 4), "ms_what": "HIP-graph replay of the forward", "nodes_per_s": round(n / tg * 1e3),
            "roofline": _small_forward_roofline(tg, bytes_, flops)}


def config_c3(dev, reps=50):
    """c3: RAGraph_graph on PROTEINS-style batches of 16 graphs, 1113-key bank, k = 3."""
    from ragraph_amd.data import DataLoader, synthetic_tu_dataset
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraphGraph

    ds = synthetic_tu_dataset(num_graphs=1113, num_node_attributes=1, num_node_labels=3, num_classes=2, seed=9)
    F3, D, N = 4, 256, 1113
    m = RAGraphGraph(PrePrompt(F3, D, "prelu", 1, 0.3).to(dev), None, F3, 2, D, device=dev).eval()
    _bank(m, N, D, 2, dev, 72)
    b = next(iter(DataLoader(ds, batch_size=16)))
    Xb = b.x.to(dev)
    ab = CSRGraph.from_edge_index_sym_normalized(b.edge_index.to(dev), Xb.shape[0])
    ptr = b.ptr.to(dev)
    _ = ab.row_normalized_values()
    n = Xb.shape[0]
    with torch.no_grad():
        f = lambda: m.forward_batch(Xb, ab, ptr)
        te, tg = event_ms(f, reps), event_ms(capture(f), reps)
    bytes_ = n * F3 * 4 + 2 * (ab.nnz * 8 + 8 * n * D) + N * D * 4 + 16 * 3 * 4 * (D + 2) + 2 * D * D * 4
    flops = 2.0 * n * F3 * D + 2 * 2.0 * ab.nnz * D + 2.0 * 16 * N * D + 2.0 * 16 * D * D
    return {"config": f"c3 RAGraph_graph, 16 PROTEINS-style graphs per pass ({n} nodes), {N}-key x {D} bank, k=3",
            "eager_ms": round(te, 4), "ms": round(tg, 4), "ms_what": "HIP-graph replay of forward_batch",
            "graphs_per_s": round(16 / tg * 1e3), "roofline": _small_forward_roofline(tg, bytes_, flops)}


def config_fewshot(dev, reps=20):
    """Few-shot node flavour: structural (position codes) + semantic retrieval on a batch of 16 graphs, 20k x 256 bank."""
    from ragraph_amd.data import DataLoader, synthetic_tu_dataset
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph_fewshot import RAGraph as RAGraphFewShot

    bf = next(iter(DataLoader(synthetic_tu_dataset(num_graphs=64, num_node_attributes=18, num_node_labels=3, seed=11), batch_size=16)))
    g = torch.Generator(device=dev).manual_seed(73)
    Xf = torch.rand(bf.x.shape[0], 18, device=dev, generator=g)
    af = CSRGraph.from_edge_index_sym_normalized(bf.edge_index.to(dev), Xf.shape[0])
    logits = torch.randn(3, 256, device=dev, generator=g)
    mf = RAGraphFewShot(PrePrompt(18, 256, "prelu", 2, 0.3).to(dev), None, logits, 256, device=dev, dataset_name="ENZYMES").eval()
    Nf, n = 20_000, Xf.shape[0]
    mf.toy_graph_base.add_resources(torch.nn.functional.normalize(torch.randn(Nf, 256, device=dev, generator=g), dim=-1),
                                    torch.randn(Nf, 256, device=dev, generator=g),
                                    torch.nn.functional.one_hot(torch.randint(0, 3, (Nf,), device=dev, generator=g), 3).float(),
                                    torch.rand(Nf, 10, device=dev, generator=g))
    anchors = torch.randint(0, n, (10,), device=dev, generator=g)
    _ = af.row_normalized_values()
    with torch.no_grad():
        t = event_ms(lambda: mf(Xf, af, logits, anchors=anchors), reps)
    bytes_ = Nf * (256 + 10) * 4 + n * Nf * 4 * 3 + n * 256 * 4 * 6
    flops = 2.0 * n * Nf * (256 + 10) + 2.0 * n * 18 * 256 + 2.0 * n * 256 * 256
    return {"config": f"few-shot RAGraph_node, 16 graphs per pass ({n} nodes), {Nf} x 256 bank + 10-d position codes, k=5",
            "ms": round(t, 4), "ms_what": "eager forward", "nodes_per_s": round(n / t * 1e3),
            "roofline": _small_forward_roofline(t, bytes_, flops)}


def _retrieval_roofline(B, N, D, ms, i8):
    """One exact top-k call of B queries against N keys: the score matrix on the matrix cores at the peak of the dtype its
    filter levels run on, or one pass over the streamed copy (D bytes per key on int8, 2 D on bf16) -- whichever is longer."""
    t = ms * 1e-3
    flops = 2.0 * B * N * D
    t_mfma = flops / ((INT8_MFMA_PEAK_TOPS if i8 else BF16_MFMA_PEAK_TFLOPS) * 1e12)
    streamed = N * D * (1 if i8 else 2) + B * D * 4
    t_hbm = streamed / (HBM_PEAK_GBS * 1e9)
    return {"bound": "hbm" if t_hbm >= t_mfma else "mfma", "mfma_dtype": "int8" if i8 else "bf16",
            "TFLOPs": round(flops / t / 1e12, 1), "streamed_GB": round(streamed / 1e9, 4),
            "GBps_streamed": round(streamed / t / 1e9, 1), "frac_of_bound": round(max(t_hbm, t_mfma) / t, 4)}


def config_c5(dev, with_generate=True):
    """c5-shaped single-GPU leg: the edge flavour's bank (4M x 64, every node's smoothed embedding) -- one slab of 4096
    queries (the reference's loop unit, modules/RAGraph.py:298) and 256 queries through the product dispatch, and generate()
    (time encoding + 3 propagation layers + retrieval of all 4M nodes + fusion) once."""
    from ragraph_amd import kernels as K
    from ragraph_amd.data import synthetic_bipartite
    from ragraph_amd.RAGraph_edge import RAGraph as RAGraphEdge

    U, I, D, k = 2_200_000, 1_800_000, 64, 10
    edges, norm, times = synthetic_bipartite(U, I, edges_per_user=10, seed=10, device=dev)

    class DSx:
        num_users, num_items = U, I
    DSx.edges, DSx.edge_norm, DSx.edge_times = edges, norm, times

    class Pre:
        def generate(self):
            g = torch.Generator(device=dev).manual_seed(3)
            return 0.1 * torch.randn(U, D, device=dev, generator=g), 0.1 * torch.randn(I, D, device=dev, generator=g)
    t0 = time.perf_counter()
    m5 = RAGraphEdge(DSx, Pre(), phase="finetune", use_RAG=True, retrieve_num=k, device=dev).eval()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    nn_ = U + I
    out = {"config": f"c5 RAGraph_edge: {nn_} nodes, {edges.shape[0]} directed edges, {nn_} x {D} bank, k={k}",
           "bank_build_s": round(t_build, 3)}
    with torch.no_grad():
        m5.generate()                                # (makes the index and its copies)
        index = m5._index
        inner = index.search_index
        Nk = inner.keys_normalized.shape[0]
        q_all = torch.cat([m5.user_embedding, m5.item_embedding], 0).detach()
        for B in (4096, 256):
            q = q_all[:B].contiguous()
            for _ in range(4):     # (the dispatch settles: overflow counts and the calls' statistics arrive one call late -- the
                index.topk(q, k)   # steady state of the edge flavour's slab loop, as in bench.py's small_batch_rates)
                torch.cuda.synchronize()
            ms = event_ms(lambda: index.topk(q, k), 20)
            cap, allowed = inner._cap_i8()
            n_i8 = K.filtered_i8_levels(B, Nk, D, k) if allowed else 0
            if cap is not None:
                cap(-1)
            rec = {"ms": round(ms, 4), "queries_per_s": round(B / ms * 1e3, 1), "searched_rows": Nk}
            rec.update(_retrieval_roofline(B, Nk, D, ms, n_i8 > 0))
            out[f"slab_B{B}"] = rec
        if with_generate:
            t5 = event_ms(lambda: m5.generate(), 2, warm=1)
            m5.use_RAG = False
            t5p = event_ms(lambda: m5.generate(), 2, warm=1)
            m5.use_RAG = True
            cap, allowed = inner._cap_i8()
            n_i8 = K.filtered_i8_levels(min(nn_, index.MAX_FILTERED_BATCH), Nk, D, k) if allowed else 0
            if cap is not None:
                cap(-1)
            rec = {"ms": round(t5, 1), "propagation_only_ms": round(t5p, 2), "retrieved_queries_per_s": round(nn_ / t5 * 1e3)}
            rec.update(_retrieval_roofline(nn_, Nk, D, t5 - t5p, n_i8 > 0))
            out["generate"] = rec
    return out, m5


def configs_block(dev, with_generate=True):
    out = {}
    out["c1"] = config_c1(dev)
    out["c3"] = config_c3(dev)
    out["fewshot"] = config_fewshot(dev)
    out["c5"], m5 = config_c5(dev, with_generate)
    return out, m5


# ---- memory -------------------------------------------------------------------------------------------------------------
def memory_block(tgb):
    """Bytes of every image of the bank the model holds on the device, against the reference's bank (keys + values + labels:
    ToyGraphBase.py:31-37)."""
    def nb(t):
        return 0 if t is None else int(t.numel() * t.element_size())

    idx = tgb._index
    inner = idx.search_index if idx is not None else None
    ref = nb(tgb.resource_keys) + nb(tgb.resource_values) + nb(tgb.resource_labels)
    rec = {"keys_fp32": nb(tgb.resource_keys), "values_fp32": nb(tgb.resource_values), "labels_fp32": nb(tgb.resource_labels),
           "keys_normalized_fp32": nb(tgb._keys_normalized)}
    if inner is not None:
        N, D = inner.keys_normalized.shape
        npad = -(-N // 256) * 256
        both = nb(inner._bf16)
        rec["bf16_image"] = min(both, (npad + 1) * D * 2)
        rec["int8_image"] = max(0, both - rec["bf16_image"])
        rec["packed_fp32"] = nb(inner._packed)
        if inner._bf16 is not None and D in (64, 128, 256):
            from ragraph_amd import kernels as K
            c = K.int8_copy_classes(inner._bf16, N)   # (kept out of `total`: not bytes)
            int8_classes = {"granule_keys": 32768 // D, "granules": c["granules"], "heavy_granules": c["heavy_granules"],
                            "cut": round(c["cut"], 5), "max_abs": round(c["max_abs"], 5), "max_dk": round(c["err"], 5),
                            "max_dk_heavy": round(c["err_heavy"], 5),
                            "what": "the int8 image's two scales (csrc/filter_common.h): granules whose largest |k_i| <= cut are "
                                    "quantised on the grid cut / 127, the heavy rest on max_abs / 127; max_dk = the measured "
                                    "largest |dk| of each class, which is what each class's proven bound uses"}
        else:
            int8_classes = None
        if idx._collapsed:
            rec["unique_rows_fp32"] = nb(inner.keys_normalized)
            rec["duplicate_groups"] = nb(idx._collapsed[1]) + nb(idx._collapsed[2])
        if idx.keys_normalized is not tgb._keys_normalized and not idx._collapsed:
            rec["padded_keys_fp32"] = nb(idx.keys_normalized)
    total = sum(v for v in rec.values())
    rec["total"] = total
    if inner is not None and int8_classes is not None:
        rec["int8_classes"] = int8_classes
    rec["reference_bank"] = ref
    rec["ratio_to_reference_bank"] = round(total / max(ref, 1), 3)
    rec["what"] = ("bytes on the device; reference_bank = keys + values + labels as the reference holds them; the normalised keys, "
                   "the bf16 and int8 images (and a packed fp32 copy when the fp32 tile kernel is used) are what the exact "
                   "top-k's speed costs in memory")
    return rec


# ---- fine-tuning steps --------------------------------------------------------------------------------------------------
def _node_step_gpu(model, feats, adj, labels, opt):
    opt.zero_grad()
    logits = model(feats, adj)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    opt.step()
    return loss


def _node_step_cpu(p, X, adj_cpu, keys, vals, labs, k, hops, labels, opt, slab, sample_rows=None):
    """RAGraph_node/finetune-rag.py:77-84 with oracle/ref_torch.py's op chain (encoder detached, preprompt.py:62).
    sample_rows: retrieve only that many of the nodes (the rest of the step runs on all of them) -- the caller extrapolates."""
    from oracle import ref_torch

    import torch.nn.functional as F

    opt.zero_grad()
    with torch.no_grad():
        h = ref_torch.gcn_layer(X, adj_cpu, p["W"], p["bias"], p["alpha"])
        t0 = time.perf_counter()
        hq = h if sample_rows is None else h[:sample_rows]
        rag_emb, rag_label, _ = ref_torch.retrieve(hq, keys, vals, labs, k, slab)
        t_ret = time.perf_counter() - t0
        if sample_rows is not None:
            reps = -(-h.shape[0] // sample_rows)
            rag_emb, rag_label = rag_emb.repeat(reps, 1)[:h.shape[0]], rag_label.repeat(reps, 1)[:h.shape[0]]
        q = ref_torch.propagate(adj_cpu, h, hops)
    hidden = q * 0.5 + rag_emb * 0.5
    dec = F.linear(F.leaky_relu(F.linear(hidden, p["fc1_w"], p["fc1_b"])), p["fc2_w"], p["fc2_b"])
    logits = torch.softmax(dec, dim=1) * 0.5 + rag_label * 0.5
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    opt.step()
    return t_ret


def finetune_node(dev, cores, shape, c2=None, cpu=True):
    """One fine-tuning step of the node flavour (RAGraph_node/finetune-rag.py:69-98: forward in train mode + cross entropy +
    backward + Adam over the decoder head -- the encoder is detached, the bank carries no gradient).
    shape = "528": a batch of 16 ENZYMES-style graphs (~530 nodes, F = 18 + 3) against a 20 000 x 256 bank, k = 4
    (BASELINE.md section 2's shape); shape = "c2": c2's model, graph and bank (passed in)."""
    from ragraph_amd.data import DataLoader, synthetic_tu_dataset
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph
    from ragraph_amd.ragraph_utils import process_tu_dataset

    if shape == "c2":
        model, feats, adj = c2
        n = feats.shape[0]
        k = model.toy_graph_base.retrieve_num
    else:
        F_in, C, D, N = 18, 3, 256, 20_000
        ds = synthetic_tu_dataset(num_graphs=16, num_node_attributes=F_in, num_node_labels=C, seed=21)
        state = torch.random.get_rng_state()
        torch.manual_seed(5)
        model = RAGraph(PrePrompt(F_in, D, "prelu", 1, 0.3).to(dev), None, F_in, C, D, finetune=True, device=dev)
        torch.random.set_rng_state(state)
        _bank(model, N, D, C, dev, 74)
        feats, adj, _ = process_tu_dataset(next(iter(DataLoader(ds, batch_size=16))), F_in, device=dev)
        n = feats.shape[0]
        k = model.toy_graph_base.retrieve_num
    C = model.num_class
    labels = torch.randint(0, C, (n,), device=dev, generator=torch.Generator(device=dev).manual_seed(75))
    was_training = model.training
    params = [p for p in model.parameters() if p.requires_grad]
    saved = [p.detach().clone() for p in params]
    model.train()
    opt = torch.optim.Adam(params, lr=1e-3)
    reps = 3 if shape == "c2" else 30
    ms = event_ms(lambda: _node_step_gpu(model, feats, adj, labels, opt), reps, warm=2)
    with torch.no_grad():
        for p, s in zip(params, saved):
            p.copy_(s)
    model.train(was_training)
    rec = {"shape": f"{n} nodes, {model.toy_graph_base.resource_keys.shape[0]} x {model.emb_size} bank, k={k}",
           "ms": round(ms, 4), "steps_per_s": round(1e3 / ms, 2), "nodes_per_s": round(n / ms * 1e3, 1),
           "what": "model.train(); logits = model(features, adj); cross_entropy; backward; Adam.step (finetune-rag.py:77-84)"}
    if cpu:
        torch.set_num_threads(cores)
        conv, dec = model.pretrain_model.gcn.convs[0], model.decoder
        fc1, fc2 = dec.layers()
        p = {"W": conv.fc.weight, "bias": conv.bias, "alpha": conv.act.weight}
        p = {k_: v.detach().cpu() for k_, v in p.items()}
        for nm, t in (("fc1_w", fc1.weight), ("fc1_b", fc1.bias), ("fc2_w", fc2.weight), ("fc2_b", fc2.bias)):
            p[nm] = t.detach().cpu().clone().requires_grad_(True)
        tgb = model.toy_graph_base
        keys, vals, labs = tgb.resource_keys.cpu(), tgb.resource_values.cpu(), tgb.resource_labels.cpu()
        adj_cpu = torch.sparse_csr_tensor(adj.rowptr.cpu(), adj.col.cpu().long(), adj.val.cpu(), (n, n))
        X, lab_cpu = feats.cpu(), labels.cpu()
        opt_c = torch.optim.Adam([p["fc1_w"], p["fc1_b"], p["fc2_w"], p["fc2_b"]], lr=1e-3)
        sample = 1024 if shape == "c2" else None
        _node_step_cpu(p, X, adj_cpu, keys, vals, labs, k, model.query_graph_hop, lab_cpu, opt_c, 1024, sample)   # warm-up
        ts, rets = [], []
        for _ in range(2 if shape == "c2" else 5):
            t0 = time.perf_counter()
            rets.append(_node_step_cpu(p, X, adj_cpu, keys, vals, labs, k, model.query_graph_hop, lab_cpu, opt_c, 1024, sample))
            ts.append(time.perf_counter() - t0)
        t, tr = sorted(ts)[len(ts) // 2], sorted(rets)[len(rets) // 2]
        if sample is not None:   # the retrieval of `sample` queries extrapolated to all n; everything else ran on all n
            t = (t - tr) + tr * (n / sample)
        rec["cpu_ms"] = round(t * 1e3, 2)
        rec["cpu_cores"] = cores
        rec["cpu_what"] = ("oracle/ref_torch.py's op chain + torch autograd + Adam on the host" +
                           (f"; retrieval timed on {sample} of the {n} queries and extrapolated" if sample else ""))
        rec["speedup_vs_cpu"] = round(t * 1e3 / ms, 1)
    return rec


def finetune_edge(dev, cores, m5, cpu=True):
    """One cal_loss step of the edge flavour at c5's shape (modules/RAGraph.py:335-355: edge dropout, forward with gradients
    through gate + 3 propagation layers + retrieval of all nodes against the 4M x 64 bank, BPR + L2 on a batch of 4096
    triples, backward, Adam).  CPU beside it: the propagation / loss / backward part on torch CPU ops over the same edges,
    the retrieval timed on ONE 256-query slab (modules/RAGraph.py:298 walks slabs) and extrapolated to every slab of a step."""
    U, I = m5.num_users, m5.num_items
    g = torch.Generator().manual_seed(76)
    batch = (torch.randint(0, U, (4096,), generator=g), torch.randint(0, I, (4096,), generator=g),
             torch.randint(0, I, (4096,), generator=g))
    params = [p for p in m5.parameters() if p.requires_grad]
    saved = [p.detach().clone() for p in params]
    m5.train()
    opt = torch.optim.Adam(params, lr=1e-3)

    def step():
        opt.zero_grad()
        loss, _ = m5.cal_loss(batch)
        loss.backward()
        opt.step()
    ms = event_ms(step, 2, warm=1)
    # the reference draws the dropout mask on the HOST (one uniform per edge, utils.py:46) and so does the default here: what
    # that costs inside the step, and the step with the mask drawn on the device (RAGraph.dropout_rng = "device")
    t0 = time.perf_counter()
    for _ in range(2):
        m5.draw_edge_mask()
    torch.cuda.synchronize()
    host_draw_ms = (time.perf_counter() - t0) / 2 * 1e3
    m5.dropout_rng = "device"
    try:
        ms_dev = event_ms(step, 2, warm=1)
    finally:
        m5.dropout_rng = "host"
    with torch.no_grad():
        for p, s in zip(params, saved):
            p.copy_(s)
    m5.eval()
    nn_ = U + I
    rec = {"shape": f"{nn_} nodes, {m5.edges.shape[0]} directed edges (half dropped per step), {nn_} x {m5.emb_size} bank, "
                    f"k={m5.retrieve_num}, 4096 BPR triples", "ms": round(ms, 1), "steps_per_s": round(1e3 / ms, 3),
           "host_mask_draw_ms": round(host_draw_ms, 1), "ms_with_device_mask": round(ms_dev, 1),
           "what": "cal_loss (edge dropout 0.5, forward, BPR + L2) + backward + Adam.step (modules/RAGraph.py:335-355); `ms` "
                   "draws the dropout mask as the reference does -- torch.rand on the HOST generator, one uniform per edge + a "
                   "copy (`host_mask_draw_ms` of the step) --, `ms_with_device_mask` draws it on the device "
                   "(RAGraph.dropout_rng = 'device')"}
    if cpu:
        torch.set_num_threads(cores)
        D = m5.emb_size
        edges = m5.edges.cpu()
        keep = (torch.rand(edges.shape[0]) + 0.5).floor().bool()
        e = edges[keep]
        w = m5.edge_norm.cpu()[keep]
        ue = m5.user_embedding.detach().cpu().clone().requires_grad_(True)
        ie = m5.item_embedding.detach().cpu().clone().requires_grad_(True)
        gw = m5.gating_weight.detach().cpu().clone().requires_grad_(True)
        gb = m5.gating_bias.detach().cpu().clone().requires_grad_(True)
        opt_c = torch.optim.Adam([ue, ie, gw, gb], lr=1e-3)
        keys = m5.resource_keys.cpu()
        vals = m5.resource_values.cpu()
        # the kept edges as a sparse CSR matrix (dst <- src, weights = edge_norm): torch.sparse.mm is the host's form of _agg's
        # gather * norm -> scatter_add (modules/RAGraph.py:232-240) that keeps the autograd tape at one dense table per layer
        A = torch.sparse_coo_tensor(torch.stack([e[:, 1], e[:, 0]]), w, (U + I, U + I)).coalesce().to_sparse_csr()
        SLAB = 256   # (the reference's 4096-query slab against 4M keys is a 65-GB score matrix: a 4-GB slab of 256 here)
        t0 = time.perf_counter()
        opt_c.zero_grad()
        x = torch.cat([ue, ie])
        x = x * torch.sigmoid(x @ gw + gb)
        res = [x]
        for _ in range(m5.num_layers):
            res.append(torch.sparse.mm(A, res[-1]))
        tot = sum(res)
        t_slab0 = time.perf_counter()
        with torch.no_grad():   # one slab of the retrieval loop, as the reference computes it (bank re-normalised per slab)
            q = res[0][:SLAB].detach()
            S = torch.matmul(torch.nn.functional.normalize(q, p=2, dim=-1), torch.nn.functional.normalize(keys, p=2, dim=-1).t())
            _, idx = torch.topk(S, m5.retrieve_num, largest=True, sorted=True)
            rag = vals[idx].mean(dim=1)
            del S
        t_slab = time.perf_counter() - t_slab0
        tot = 0.7 * tot
        uo, io = tot[:U], tot[U:]
        us, ps, ns = batch
        pos, neg = (uo[us] * io[ps]).sum(1), (uo[us] * io[ns]).sum(1)
        loss = (-torch.log(1e-10 + torch.sigmoid(pos - neg))).mean()
        loss.backward()
        opt_c.step()
        t_rest = time.perf_counter() - t0 - t_slab
        slabs = -(-nn_ // SLAB)
        rec["cpu_ms"] = round((t_rest + t_slab * slabs) * 1e3, 1)
        rec["cpu_cores"] = cores
        rec["cpu_parts"] = {"propagation_loss_backward_adam_s": round(t_rest, 2), "one_retrieval_slab_s": round(t_slab, 2), "slabs": slabs}
        rec["cpu_what"] = ("torch-CPU restatement of the step (one cold run): gate + 3 torch.sparse.mm layers + BPR + backward + Adam "
                           "measured in full, the retrieval on ONE slab of 256 queries x the 4M x 64 bank (a 4-GB score slab; bank "
                           "re-normalised per slab as the reference does) extrapolated to every slab of the step")
        rec["speedup_vs_cpu"] = round(rec["cpu_ms"] / ms, 1)
    return rec
