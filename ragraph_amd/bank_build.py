"""Toy-bank construction (the step BEFORE the hot path) -- ToyGraphBase.py:40-45,91-119.

Encoder inference, key normalisation and value propagation run on the HIP kernels.  The stochastic parts
(feature noise / node drop / edge rewrite, PageRank + degree inverse-importance sampling: Augmentation.py:8-64,
InverseSampling.py:6-56) are small per-graph tensor bookkeeping (n ~ 40) done with torch ops on the device; they draw
from torch's RNG like the reference, so a bank is reproducible per seed but not bit-identical to a CPU-reference bank.
Position-aware codes (Floyd-Warshall, PositionAwareEncoder.py) are only consumed by the few-shot variant and are not
built (section 8f row 3).
"""
from __future__ import annotations

import torch

from . import kernels as K
from .data import DataLoader
from .graph import CSRGraph
from .ragraph_utils.Propagation import Propagation
from .ragraph_utils.utility import process_tu_dataset


def _dense(g: CSRGraph) -> torch.Tensor:
    a = torch.zeros(g.n, g.n, device=g.device)
    rows = torch.repeat_interleave(torch.arange(g.n, device=g.device), g.rowptr[1:] - g.rowptr[:-1])
    a[rows, g.col.long()] = g.val
    return a


def compute_sample_prob(adj: torch.Tensor) -> torch.Tensor:
    """InverseSampling.py:6-56 on a small dense adjacency: p ~ 1 / (0.5*PageRank + 0.5*degree_centrality + 1e-6)."""
    n = adj.shape[0]
    out_deg = adj.sum(dim=1)
    zero = out_deg == 0
    out_deg = torch.where(zero, torch.ones_like(out_deg), out_deg)
    P = adj / out_deg[:, None]
    P[zero] = 1.0 / n
    Pt = P.t().contiguous()
    p = torch.full((n,), 1.0 / n, device=adj.device)
    for _ in range(125):  # convergence is tested once per 8 power iterations: one host sync instead of eight
        done = None
        for _ in range(8):
            new_p = (1 - 0.85) / n + 0.85 * torch.mv(Pt, p)
            done = torch.norm(new_p - p, p=1) < 1e-6
            p = new_p
        if bool(done):
            break
    dc = adj.sum(dim=0) / max(n - 1, 1)
    inv = 1.0 / (0.5 * p + 0.5 * dc + 1e-6)
    return inv / inv.sum()


def augment_graph(num_augment_scale, features, adj_dense):
    """Augmentation.py:51-64: the original graph, then `num_augment_scale` noisy / dropped / rewired copies."""
    prob = compute_sample_prob(adj_dense)
    yield features, adj_dense
    for _ in range(num_augment_scale):
        noisy = features + torch.randn_like(features) * 0.1
        mask = torch.bernoulli(prob * 0.01).unsqueeze(-1)                      # Augmentation.py:17-18
        keep = (prob.unsqueeze(1) + prob.unsqueeze(0)) / 2
        new_adj = (torch.rand_like(adj_dense) < keep).float()                  # Augmentation.py:23-27
        yield noisy * mask, new_adj


def build_toy_graph(tgb, resource_dataset) -> None:
    """ToyGraphBase.build_toy_graph: one resource graph at a time (DataLoader batch_size=1, :42)."""
    dev = tgb.device
    for data in DataLoader(resource_dataset, batch_size=1, shuffle=False):
        features, adj, node_labels = process_tu_dataset(data, resource_dataset.num_node_attributes, device=dev)
        graph_label = None
        if tgb.flavour == "graph":
            graph_label = torch.nn.functional.one_hot(data.y.reshape(-1)[:1].to(dev).long(),
                                                      tgb.resource_labels.shape[1]).float()
        _build_one(tgb, features, adj, node_labels, graph_label)


def _build_one(tgb, features, adj: CSRGraph, node_labels, graph_label):
    """_build_toy_graph_base, ToyGraphBase.py:91-119 (node) / RAGraph_graph/...:99-127 (graph)."""
    adj_dense = _dense(adj) if (tgb.num_augment_scale > 0 or tgb.num_inverse_sample > 0) else None
    if adj_dense is None:
        variants = [(features, adj)]
    else:
        variants = list(augment_graph(tgb.num_augment_scale, features, adj_dense))
    for aug_features, aug_adj in variants:
        emb = tgb.pretrain_model.inference(aug_features, aug_adj)                              # :93
        if tgb.num_inverse_sample > 0:
            a = aug_adj if isinstance(aug_adj, torch.Tensor) else adj_dense
            prob = compute_sample_prob(a)                                                      # :97
            pick = torch.multinomial(prob, num_samples=tgb.num_inverse_sample, replacement=True)  # :98
            sample_adj = adj_dense[pick, :][:, pick]                                           # :100 (ORIGINAL adj)
            keys, labels = emb[pick], node_labels[pick]
        else:
            sample_adj, keys, labels = aug_adj, emb, node_labels
        keys = K.normalize_rows(keys)                                                          # :109
        values = Propagation.aggregate_k_hop_features(sample_adj, keys, tgb.toy_graph_hop)    # :112
        if tgb.flavour == "graph":                                                             # graph :115-121
            seg = torch.tensor([0, keys.shape[0]], dtype=torch.int64, device=keys.device)
            keys = K.segment_reduce(keys, seg, mean_mode=True)
            values = K.segment_reduce(values, seg, mean_mode=True)
            labels = graph_label
        tgb.add_resources(keys, values, labels)                                                # :116-119
