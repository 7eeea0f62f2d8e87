"""PyTorch-CPU restatement of the reference's hot path, op for op AS THE REFERENCE COMPUTES IT.  TEST INFRASTRUCTURE
ONLY: it is the timed `cpu_baseline` ("port") of bench.py and a second, independent check of the golden vectors.

Unlike oracle/ragraph_oracle.c (which fixes one summation order so the GPU can be compared bit for bit), this file
keeps the reference's own op chain -- F.normalize on both operands on every call, a materialised B x N score slab,
torch.topk, fancy-index gathers -- so its speed is what a user of the reference gets on these host cores.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def cosine_similarity(search_keys, resource_keys):
    """RAGraph_node/ragraph_utils/SimilarityFunctions.py:6-16."""
    return torch.matmul(F.normalize(search_keys, p=2, dim=-1), F.normalize(resource_keys, p=2, dim=-1).t())


def retrieve(search_keys, keys, values, labels, k, slab=1024, renormalize_bank=True):
    """ToyGraphBase.py:47-81 looped over query slabs as RAGraph_edge/modules/RAGraph.py:298-324 does (the un-slabbed
    score matrix would not fit at 1M keys).  renormalize_bank=False is the 'fair CPU' variant (bank normalised once)."""
    out_e, out_l, out_i = [], [], []
    kn = None if renormalize_bank else F.normalize(keys, p=2, dim=-1)
    for s in range(0, search_keys.shape[0], slab):
        q = search_keys[s:s + slab]
        if renormalize_bank:
            S = cosine_similarity(q, keys)
        else:
            S = torch.matmul(F.normalize(q, p=2, dim=-1), kn.t())
        _, idx = torch.topk(S, k, largest=True, sorted=True)
        out_i.append(idx)
        out_e.append(values[idx].sum(dim=1))
        if labels is not None:
            out_l.append(labels[idx].mean(dim=1))
    return torch.cat(out_e), (torch.cat(out_l) if out_l else None), torch.cat(out_i)


def gcn_layer(X, adj, W, bias, alpha):
    """layers/gcn.py:26-40; adj dense (the reference's form) or torch sparse CSR (needed beyond n ~ 16k)."""
    fts = X @ W.t()
    out = torch.sparse.mm(adj, fts) if adj.layout != torch.strided else torch.mm(adj, fts)
    return F.prelu(out + bias, alpha.reshape(1))


def propagate(adj, x, k):
    """Propagation.py:7-27 (dense), or the same on a row-normalised sparse CSR matrix."""
    if adj.layout == torch.strided:
        adjn = adj / adj.sum(dim=1, keepdim=True)
        for _ in range(k):
            x = F.relu(adjn @ x)
        return x
    crow, col, val = adj.crow_indices(), adj.col_indices(), adj.values()
    rows = torch.repeat_interleave(torch.arange(adj.shape[0]), crow[1:] - crow[:-1])
    deg = torch.zeros(adj.shape[0]).index_add_(0, rows, val)
    adjn = torch.sparse_csr_tensor(crow, col, val / deg[rows], adj.shape)
    for _ in range(k):
        x = F.relu(torch.sparse.mm(adjn, x))
    return x


def node_forward(X, adj, p, keys, values, labels, k, hops, retrieve_weight, label_weight, slab=1024,
                 renormalize_bank=True):
    """RAGraph_node/RAGraph.py:39-59."""
    h = gcn_layer(X, adj, p["W"], p["bias"], p["alpha"])
    rag_emb, rag_label, idx = retrieve(h, keys, values, labels, k, slab, renormalize_bank)
    q = propagate(adj, h, hops)
    hidden = q * (1 - retrieve_weight) + rag_emb * retrieve_weight
    dec = F.linear(F.leaky_relu(F.linear(hidden, p["fc1_w"], p["fc1_b"])), p["fc2_w"], p["fc2_b"])
    return torch.softmax(dec, dim=1) * (1 - label_weight) + rag_label * label_weight, idx
