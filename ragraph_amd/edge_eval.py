"""Top-k recommendation evaluation of RAGraph_edge (RAGraph_edge/utils/metrics.py:83-141): generate() once, then per
batch of users: rating = user_emb @ item_emb.T, history items masked to -1e8, top-k.  The reference moves every rating
slab to the CPU for the mask loop and torch.topk; here the slab stays in HBM (linear -> scatter_fill -> topk_rows)."""
from __future__ import annotations

import numpy as np
import torch

from . import kernels as K


@torch.no_grad()
def topk_items(model, users: torch.Tensor, hist_rowptr: torch.Tensor, hist_items: torch.Tensor, k: int = 20,
               eval_batch_size: int = 512, embeddings=None):
    """-> idx [len(users), k] item ids ranked by rating.  hist_rowptr/hist_items: CSR over `users` (in that order) of
    the training-history item ids to exclude (metrics.py:210-214)."""
    user_emb, item_emb = embeddings if embeddings is not None else model.generate()
    out = []
    for s in range(0, users.numel(), eval_batch_size):
        ub = users[s:s + eval_batch_size]
        rating = K.linear(K.gather_rows(user_emb, ub), item_emb)                   # metrics.py:112 model.rating
        rp = hist_rowptr[s:s + ub.numel() + 1] - hist_rowptr[s]
        cols = hist_items[int(hist_rowptr[s]):int(hist_rowptr[s + ub.numel()])]
        if cols.numel():
            K.scatter_fill_(rating, rp.contiguous(), cols.contiguous(), -1e8)       # metrics.py:114 _mask_history_pos
        out.append(K.topk_rows(rating, k)[1])                                       # metrics.py:116
    return torch.cat(out)


def recall_ndcg(rank_idx: np.ndarray, ground_truth: list, k: int = 20):
    """metrics.py:12-46 (recall@k, ndcg@k), host side on the [U,k] index matrix."""
    hits = np.zeros(rank_idx.shape, dtype=np.float64)
    for u, items in enumerate(ground_truth):
        hits[u] = np.isin(rank_idx[u], list(items))
    n_rel = np.array([len(g) for g in ground_truth], dtype=np.float64)
    recall = float(np.sum(hits[:, :k].sum(1) / n_rel))
    disc = 1.0 / np.log2(np.arange(2, k + 2))
    ideal = np.array([disc[:min(k, int(n))].sum() for n in n_rel])
    ideal[ideal == 0] = 1.0
    ndcg = float(np.sum((hits[:, :k] * disc).sum(1) / ideal))
    return recall / len(ground_truth), ndcg / len(ground_truth)
