"""Toy-bank construction on the device, batched over the resource graphs (the step BEFORE the hot path; SURVEY.md
section 8f row 1) -- ToyGraphBase.py:40-45,91-119 (node), RAGraph_graph/ragraph_utils/ToyGraphBase.py:99-127 (graph),
RAGraph_edge/modules/RAGraph.py:185-226 (edge: ragraph_amd/RAGraph_edge.py calls compute_sample_prob).

The reference walks the resource graphs one by one in Python (DataLoader batch_size = 1): per graph a dense PageRank with a
host-side convergence test per power iteration, a Python double loop for the position codes, and a torch.cat of the whole
bank.  Here ALL resource graphs form one block-diagonal batch and every arithmetic step is one batched HIP launch:

  encoder inference                        linear + spmm_csr over the block-diagonal CSR            (K3)
  InverseSampling.compute_sample_prob      csr_row_sums + pagerank (all power iterations enqueued, no host round trip)
                                           + sample_prob, one segment per graph                     InverseSampling.py:6-56
  key normalisation / value propagation    normalize_rows, spmm_csr over the sampled toy graphs     ToyGraphBase.py:109-112
  position-aware codes                     position_codes_batch (Floyd-Warshall in LDS per graph)   PositionAwareEncoder.py:6-48
  bank append                              amortised _Bank growth (no quadratic torch.cat)          ToyGraphBase.py:116-119

The STOCHASTIC draws (feature noise, node drop, edge rewrite: Augmentation.py:8-29; multinomial inverse-importance
sampling: ToyGraphBase.py:98; anchors: PositionAwareEncoder.py:11) come from torch's RNG, batched over the graphs; the
reference draws them from the CUDA generator per graph, so a bank is reproducible per seed here but cannot be
draw-for-draw identical to a reference bank (the deterministic parts are pinned by golden g13 and the oracle).
"""
from __future__ import annotations

import torch

from . import kernels as K
from .data import DataLoader
from .graph import CSRGraph
from .ragraph_utils.Propagation import Propagation
from .ragraph_utils.utility import process_tu_dataset

NUM_ANCHORS, DIS_Q = 10, 10.0   # ToyGraphBase.py:27-28 (num_anchors, dis_q)


PAGERANK_MAX_ITER = 128   # d = 0.85, eps = 1e-6 converge within ~90 power iterations


def compute_sample_prob(adj, graph_ptr: torch.Tensor | None = None, check: bool = True) -> torch.Tensor:
    """InverseSampling.compute_sample_prob (InverseSampling.py:6-19) for every graph of a block-diagonal batch at once:
    p ~ 1 / (0.5 * PageRank + 0.5 * degree centrality + 1e-6), normalised per graph.  `adj`: CSRGraph (or a dense
    tensor); graph_ptr [G+1] node offsets (default: one graph).  The reference iterates PageRank until it converges
    (`while True`, InverseSampling.py:38-44); here PAGERANK_MAX_ITER iterations are enqueued up front, and `check` reads
    the per-graph iteration counts back (one synchronisation: bank construction, not the hot path) and raises if a graph
    used them all -- an unconverged iterate is never returned silently.  check=False: no host synchronisation."""
    g = adj if isinstance(adj, CSRGraph) else CSRGraph.from_dense(adj)
    if graph_ptr is None:
        graph_ptr = torch.tensor([0, g.n], dtype=torch.int64, device=g.device)
    gt = g.transposed()
    out_deg = K.csr_row_sums(g.rowptr, g.val)                    # :25 torch.sum(adj, dim=1)
    p, iters = K.pagerank(gt.rowptr, gt.col, gt.val, out_deg, graph_ptr, max_iter=PAGERANK_MAX_ITER)   # :22-47
    col_sum = K.csr_row_sums(gt.rowptr, gt.val)                  # :53 torch.sum(adj, dim=0)
    prob = K.sample_prob(p, col_sum, graph_ptr)                  # :10-17
    if check and int(iters.max()) >= PAGERANK_MAX_ITER:
        raise RuntimeError(f"PageRank did not converge within {PAGERANK_MAX_ITER} iterations for "
                           f"{int((iters >= PAGERANK_MAX_ITER).sum())} of {iters.numel()} graphs (the reference iterates "
                           f"until ||dp||_1 < 1e-6, InverseSampling.py:38-44)")
    return prob


def _intra_graph_pairs(graph_ptr: torch.Tensor):
    """All ordered node pairs (i, j) inside each graph of the batch, as global ids."""
    sizes = graph_ptr[1:] - graph_ptr[:-1]
    sq = sizes * sizes
    gid = torch.repeat_interleave(torch.arange(sizes.numel(), device=sizes.device), sq)
    start = torch.cumsum(sq, 0) - sq
    local = torch.arange(int(sq.sum()), device=sizes.device) - start[gid]
    n_g = sizes[gid]
    return graph_ptr[gid] + local // n_g, graph_ptr[gid] + local % n_g


def augment_batch(features: torch.Tensor, prob: torch.Tensor, graph_ptr: torch.Tensor):
    """One augmented copy of every graph of the batch (Augmentation.py:8-29): Gaussian feature noise (sigma 0.1), node drop
    with probability mask Bernoulli(prob * 0.01), every edge slot rewritten to 1 with probability (p_i + p_j) / 2."""
    noisy = features + torch.randn_like(features) * 0.1
    mask = torch.bernoulli(prob * 0.01).unsqueeze(-1)                           # :17-18
    i, j = _intra_graph_pairs(graph_ptr)
    keep = torch.rand(i.shape, device=i.device) < (prob[i] + prob[j]) / 2       # :23-27
    i, j = i[keep], j[keep]
    g, _ = CSRGraph.from_coo(i, j, torch.ones(i.shape, device=i.device), features.shape[0], sort_cols=True)
    return noisy * mask, g


def _dense_blocks(g: CSRGraph, graph_ptr: torch.Tensor, pick: torch.Tensor) -> torch.Tensor:
    """adj[pick_g][:, pick_g] of every graph (ToyGraphBase.py:100): pick [G,S] global node ids -> [G,S,S] dense."""
    G, S = pick.shape
    rows = g.row_ids()
    key = rows * g.n + g.col.long()                                             # sorted (CSR order, ascending columns)
    want = (pick.unsqueeze(2) * g.n + pick.unsqueeze(1)).reshape(-1)
    pos = torch.searchsorted(key, want).clamp_(max=max(key.numel() - 1, 0))
    hit = key[pos] == want if key.numel() else torch.zeros_like(want, dtype=torch.bool)
    return torch.where(hit, g.val[pos], torch.zeros((), device=g.device)).reshape(G, S, S)


def build_toy_graph(tgb, resource_dataset, batch_size: int = 4096) -> None:
    """ToyGraphBase.build_toy_graph (ToyGraphBase.py:40-45): the reference appends one resource graph at a time
    (DataLoader batch_size = 1); here whole batches of graphs go through each kernel at once."""
    dev = tgb.device
    for data in DataLoader(resource_dataset, batch_size=batch_size, shuffle=False):
        features, adj, node_labels = process_tu_dataset(data, resource_dataset.num_node_attributes, device=dev)
        graph_ptr = data.ptr.to(dev, torch.int64)
        graph_labels = None
        if tgb.flavour in ("graph", "graph_fewshot"):
            graph_labels = torch.nn.functional.one_hot(data.y.reshape(-1).to(dev).long(),
                                                       tgb.resource_labels.shape[1]).float()
        _build_batch(tgb, features, adj, node_labels, graph_ptr, graph_labels)


def _build_batch(tgb, features, adj: CSRGraph, node_labels, graph_ptr, graph_labels):
    """_build_toy_graph_base for a batch of graphs (ToyGraphBase.py:91-119 node; RAGraph_graph/...:99-127 graph)."""
    G = graph_ptr.numel() - 1
    S = tgb.num_inverse_sample
    variants = [(features, adj)]
    if tgb.num_augment_scale > 0:                                               # Augmentation.augment_graph :51-64
        prob0 = compute_sample_prob(adj, graph_ptr)
        variants += [augment_batch(features, prob0, graph_ptr) for _ in range(tgb.num_augment_scale)]
    embed = tgb.pretrain_model.encode if tgb.flavour == "graph_fewshot" else tgb.pretrain_model.inference
    for aug_features, aug_adj in variants:
        emb = embed(aug_features, aug_adj)                                      # :93 (graph_fewshot :120 encode)
        if S > 0:
            prob = compute_sample_prob(aug_adj, graph_ptr)                      # :97
            sizes = graph_ptr[1:] - graph_ptr[:-1]
            maxn = int(sizes.max())
            local = torch.arange(maxn, device=prob.device).unsqueeze(0).expand(G, maxn)
            valid = local < sizes.unsqueeze(1)
            padded = torch.zeros((G, maxn), device=prob.device)
            padded[valid] = prob
            pick = graph_ptr[:-1].unsqueeze(1) + torch.multinomial(padded, S, replacement=True)   # :98, all graphs at once
            blocks = _dense_blocks(adj, graph_ptr, pick)                        # :100 sample_adj from the ORIGINAL adj
            flat = pick.reshape(-1)
            keys = K.normalize_rows(K.gather_rows(emb, flat))                   # :101,109
            labels = K.gather_rows(node_labels, flat)
            sample_csr = CSRGraph.from_dense(torch.block_diag(*blocks)) if G * S <= 4096 else _blocks_to_csr(blocks)
            values = Propagation.aggregate_k_hop_features(sample_csr, keys, tgb.toy_graph_hop)    # :112
            anchors = torch.randint(low=0, high=S, size=(G, NUM_ANCHORS)).to(prob.device)          # PositionAwareEncoder.py:11
            positions = K.position_codes_batch(blocks, anchors, DIS_Q).reshape(G * S, NUM_ANCHORS)  # :114
            seg_ptr = torch.arange(0, G * S + 1, S, dtype=torch.int64, device=prob.device)
        else:
            keys = K.normalize_rows(emb)
            values = Propagation.aggregate_k_hop_features(aug_adj, keys, tgb.toy_graph_hop)
            labels, positions, seg_ptr = node_labels, None, graph_ptr
        if tgb.flavour == "graph":                                              # graph :115-121: one row per graph
            keys = K.segment_reduce(keys, seg_ptr, mean_mode=True)
            values = K.segment_reduce(values, seg_ptr, mean_mode=True)
            labels, positions = graph_labels, None
        elif tgb.flavour == "graph_fewshot":   # RAGraph_graph_fewshot/.../ToyGraphBase.py:118-126: every node, its graph's label
            rows = seg_ptr[1:] - seg_ptr[:-1]
            labels, positions = K.gather_rows(graph_labels, torch.repeat_interleave(
                torch.arange(rows.numel(), device=rows.device), rows)), None
        tgb.add_resources(keys, values, labels, positions=positions)            # :116-119


def _blocks_to_csr(blocks: torch.Tensor) -> CSRGraph:
    """[G,S,S] dense blocks -> block-diagonal CSR without materialising the (G S)^2 matrix."""
    G, S, _ = blocks.shape
    nz = torch.nonzero(blocks, as_tuple=False)                                  # (g, a, b) in row-major order
    rows, cols = nz[:, 0] * S + nz[:, 1], nz[:, 0] * S + nz[:, 2]
    rowptr = torch.zeros(G * S + 1, dtype=torch.int64, device=blocks.device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=G * S), 0)
    return CSRGraph(rowptr, cols.to(torch.int32), blocks[nz[:, 0], nz[:, 1], nz[:, 2]].contiguous(), G * S)


def build_reference_recipe_bank(pretrain_model, n_rows: int, num_node_attributes: int, num_class: int, emb_size: int,
                                query_graph_hop: int = 3, seed: int = 21, device="cuda", attr_dist: str = "normal"):
    """A node-flavour bank of n_rows rows made by the reference's OWN recipe (ToyGraphBase.py:40-45,91-119 through
    build_toy_graph above) over synthetic resource graphs: per graph one original pass + num_augment_scale = 3 augmented
    passes of num_inverse_sample = 10 rows drawn with replacement.  The augmented passes see features multiplied by
    bernoulli(sample_prob * 0.01) (Augmentation.py:17-18) -- zero for practically every node -- so their rows are all
    normalize(PReLU(bias)): three quarters of such a bank are copies of one vector, and the sampled rows repeat as well.
    Measurement and test input (bench.py `retrieval_reference_bank`, tests/test_gpu_fullsize.py); returns the ToyGraphBase."""
    from .data import synthetic_tu_dataset
    from .ragraph_utils.ToyGraphBase import ToyGraphBase

    tgb = ToyGraphBase(pretrain_model, num_class, emb_size, query_graph_hop, device=device, flavour="node")
    per_graph = (1 + tgb.num_augment_scale) * tgb.num_inverse_sample
    ds = synthetic_tu_dataset(num_graphs=-(-n_rows // per_graph), num_node_attributes=num_node_attributes,
                              num_node_labels=num_class, seed=seed, attr_dist=attr_dist)
    build_toy_graph(tgb, ds)
    return tgb
