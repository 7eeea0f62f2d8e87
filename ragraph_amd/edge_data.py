"""Edge-list ingestion for the link-prediction flavour -- a vectorised restatement of
RAGraph_edge/utils/dataloader.py:14-124,186-196 and RAGraph_edge/modules/base_model.py:34-52.

File format (the reference's): one line per user, TAB-separated `user \t items (space-sep) \t unix times (space-sep)`.
Produces what RAGraph_edge.RAGraph consumes, directly as device tensors:
  edges [2E,2] int64 (src,dst) over the joint id space (items offset by num_users), both directions, sorted by
  destination then source (the order the reference's transposed scipy product leaves them in), edge_norm [2E] = d^-1/2[src] d^-1/2[dst] on the 0/1 bipartite
  graph, edge_times [2E] = 1 + (t - t_min) // (hour_interval * 3600)  (dataloader.py:94,186-196).
The reference builds these through Python loops over every edge and a dict-of-dicts; here it is numpy on the host
(one-off preprocessing, not on the per-forward path).
"""
from __future__ import annotations

import numpy as np
import torch


def _read(path, has_time=True):
    users, items, times = [], [], []
    per_user = {}
    with open(path, "r") as f:
        for line in f:
            parts = line.rstrip("\n").split("\t")
            if len(parts) < 2 or not parts[1]:
                continue
            u = int(parts[0])
            it = np.array(parts[1].split(" "), dtype=np.int64)
            tm = np.array(parts[2].split(" "), dtype=np.int64) if (has_time and len(parts) > 2) else np.zeros_like(it)
            users.append(np.full(it.shape, u, dtype=np.int64))
            items.append(it)
            times.append(tm)
            per_user[u] = it.tolist()
    cat = (lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int64))
    return cat(users), cat(items), cat(times), per_user


class EdgeListData:
    def __init__(self, train_file, test_file=None, hour_interval=1, has_time=True, num_users=None, num_items=None,
                 device="cuda"):
        u, i, t, self.train_user_dict = _read(train_file, has_time)
        self.test_user_dict = _read(test_file, False)[3] if test_file else {}
        tu = max(self.test_user_dict) + 1 if self.test_user_dict else 0
        ti = max(max(v) for v in self.test_user_dict.values()) + 1 if self.test_user_dict else 0
        self.num_users = int(num_users or max(u.max() + 1, tu))          # dataloader.py:100-101
        self.num_items = int(num_items or max(i.max() + 1, ti))
        self.num_edges = int(u.shape[0])
        step = 1 + (t - t.min()) // int(hour_interval * 3600)            # :94,186-196 (0 is the self-loop padding)
        self.user_hist_dict = {uu: self.train_user_dict.get(uu, []) for uu in range(self.num_users)}
        if torch.device(device).type == "cuda":   # the product path: sorts, degrees and norms on the device (csrc/ingest.hip)
            from . import kernels as K
            self.edges, self.edge_norm, self.edge_times = K.binorm_edges(
                torch.from_numpy(u).to(device), torch.from_numpy(i).to(device), torch.from_numpy(step.astype(np.int64)).to(device),
                self.num_users, self.num_items)
        else:                                      # host tensors (the CPU host-logic test): the numpy restatement below
            edges, norm, times = binorm_edges(self.num_users, self.num_items, u, i, step)
            self.edges = torch.from_numpy(edges).to(device)
            self.edge_norm = torch.from_numpy(norm).to(device)
            self.edge_times = torch.from_numpy(times).to(device)

    def history_csr(self, users, device="cuda"):
        """CSR of training-history items for `users` (the mask of metrics.py:210-214)."""
        lens = [len(self.user_hist_dict.get(int(x), [])) for x in users]
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        cols = (np.concatenate([np.asarray(self.user_hist_dict.get(int(x), []), dtype=np.int64) for x in users])
                if sum(lens) else np.zeros(0, np.int64))
        return torch.from_numpy(rowptr).to(device), torch.from_numpy(cols).to(device)


def binorm_edges(num_users, num_items, u, i, step):
    """base_model.py:34-52: symmetric bipartite adjacency, binarised, D^-1/2 A D^-1/2, as a sorted COO edge list; the
    time of a duplicated (user,item) pair is its LAST occurrence (the dict assignment of dataloader.py:112-113)."""
    n = num_users + num_items
    key = u * num_items + i
    order = np.argsort(key, kind="stable")
    ks = key[order]
    last = np.r_[ks[1:] != ks[:-1], True] if ks.size else np.zeros(0, bool)   # last occurrence of each pair
    uu, ii, tt = u[order][last], i[order][last] + num_users, step[order][last]
    src = np.concatenate([uu, ii])
    dst = np.concatenate([ii, uu])
    tim = np.concatenate([tt, tt])
    deg = np.bincount(src, minlength=n).astype(np.float64)
    with np.errstate(divide="ignore"):
        dinv = np.power(deg, -0.5)
    dinv[np.isinf(dinv)] = 0.0
    norm = (dinv[src] * dinv[dst]).astype(np.float32)   # float64 product cast to fp32, as mat.data.astype(np.float32)
    o = np.lexsort((src, dst))   # by destination, then source: the order (A D)^T D .tocoo() leaves the reference
    return np.stack([src[o], dst[o]], 1).astype(np.int64), norm[o], tim[o].astype(np.int64)
