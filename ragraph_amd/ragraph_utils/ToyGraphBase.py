"""Mirror of RAGraph_node/ragraph_utils/ToyGraphBase.py (and the graph flavour): the toy-graph vector library.

Bank layout in HBM: keys [N,D] fp32 (unit rows, as stored by the reference, ToyGraphBase.py:109), values [N,D],
labels [N,C] one-hot fp32, plus `keys_normalized` -- F.normalize(keys) computed ONCE per bank version; the reference
recomputes it on every retrieve (SimilarityFunctions.py:11; 2 GB of traffic per call at 1M x 256).
Storage grows geometrically (the reference re-allocates with torch.cat per resource graph, ToyGraphBase.py:116-119).
"""
from __future__ import annotations

import torch
from torch import Tensor

from .. import kernels as K
from .Propagation import Propagation


class _Bank:
    """Append-only row store with amortised growth."""

    def __init__(self, width: int, device):
        self.buf = torch.empty((0, width), dtype=torch.float32, device=device)
        self.n = 0

    def append(self, rows: Tensor):
        rows = rows.to(self.buf.device, torch.float32)
        need = self.n + rows.shape[0]
        if need > self.buf.shape[0]:
            new = torch.empty((max(need, 2 * self.buf.shape[0], 1024), self.buf.shape[1]), dtype=torch.float32,
                              device=self.buf.device)
            new[:self.n] = self.buf[:self.n]
            self.buf = new
        self.buf[self.n:need] = rows
        self.n = need

    def view(self) -> Tensor:
        return self.buf[:self.n]


class ToyGraphBase:
    def __init__(self, pretrain_model, num_class, emb_size, query_graph_hop, device="cuda", flavour="node") -> None:
        self.flavour = flavour
        if flavour == "node":   # RAGraph_node/ragraph_utils/ToyGraphBase.py:18-29
            self.num_inverse_sample = 10
            self.num_augment_scale = 3
            self.retrieve_num = num_class + 1
        else:                   # RAGraph_graph/ragraph_utils/ToyGraphBase.py:21-27 (graph_fewshot: the same constants)
            self.num_inverse_sample = 0
            self.num_augment_scale = 0
            self.retrieve_num = min(3, num_class + 1)
        self.noise_retrieve_num = 1
        self.noise_std = 0.01
        self.toy_graph_hop = query_graph_hop - 1
        self.pretrain_model = pretrain_model
        self.device = torch.device(device)
        self._keys = _Bank(emb_size, self.device)
        self._values = _Bank(emb_size, self.device)
        self._labels = _Bank(num_class, self.device)
        self.num_anchors, self.dis_q = 10, 10             # ToyGraphBase.py:27-28
        self._positions = _Bank(self.num_anchors, self.device)   # position-aware codes of the sampled toy graphs (:114,119)
        self._keys_normalized = None  # cache, invalidated by every append
        self._index = None            # K.KeyIndex of this bank version (packed / bf16 copies made on first use)

    # ---- bank state (attribute names of the reference) ---------------------------------------------------------
    @property
    def resource_keys(self) -> Tensor:
        return self._keys.view()

    @property
    def resource_values(self) -> Tensor:
        return self._values.view()

    @property
    def resource_labels(self) -> Tensor:
        return self._labels.view()

    @property
    def resource_positions(self) -> Tensor:
        return self._positions.view()

    def add_resources(self, keys: Tensor, values: Tensor, labels: Tensor, positions: Tensor | None = None) -> None:
        """Append rows to the bank (what ToyGraphBase.py:116-119 does with torch.cat)."""
        assert keys.shape[0] == values.shape[0] == labels.shape[0]
        self._keys.append(keys)
        self._values.append(values)
        self._labels.append(labels)
        if positions is not None:
            assert positions.shape[0] == keys.shape[0]
            self._positions.append(positions)
        self._keys_normalized = self._index = None

    def set_resources(self, keys: Tensor, values: Tensor, labels: Tensor) -> None:
        """Adopt caller-owned device tensors as the bank without copying (e.g. a 1M-row synthetic bank)."""
        for b, t in ((self._keys, keys), (self._values, values), (self._labels, labels)):
            b.buf, b.n = t.to(self.device, torch.float32).contiguous(), t.shape[0]
        self._keys_normalized = self._index = None

    @property
    def keys_normalized(self) -> Tensor:
        if self._keys_normalized is None:
            self._keys_normalized = K.normalize_rows(self.resource_keys)
        return self._keys_normalized

    # ---- build (the step before the hot path; deterministic part) -----------------------------------------------
    def build_toy_graph(self, resource_dataset):
        """ToyGraphBase.py:40-45.  One resource graph per batch; the stochastic augmentation + inverse-importance
        sampling of the node flavour (ToyGraphBase.py:92-102) lives in ragraph_amd.bank_build."""
        from ..bank_build import build_toy_graph
        build_toy_graph(self, resource_dataset)

    # ---- retrieve (hot path) -----------------------------------------------------------------------------------
    def topk(self, search_keys: Tensor, k: int):
        """(scores [B,k], idx [B,k]) of the fused cosine + top-k kernel; canonical tie order."""
        q = search_keys.reshape(1, -1) if search_keys.dim() == 1 else search_keys
        if self.resource_keys.shape[0] < k:
            raise RuntimeError(f"selected index k out of range: bank has {self.resource_keys.shape[0]} rows, k={k}")
        if self._index is None:
            self._index = K.KeyIndex(self.keys_normalized)
        return self._index.topk(q, k)  # fp32 streaming / tile kernel or, for large batches, the bf16-filtered exact path

    def retrieve_indices(self, search_keys: Tensor, add_noise: bool) -> Tensor:
        """The rows retrieve() gathers: the top-k' indices (k' = 2 * retrieve_num with add_noise, :66) and, in the node
        flavour with add_noise, noise_retrieve_num uniformly random rows behind them (:73-79).  The reference draws the
        noise from torch's default CPU generator (no device argument) and only then moves it to the bank's device: drawn
        the same way here, so torch.manual_seed reproduces the reference's rows."""
        retrieve_num = 2 * self.retrieve_num if add_noise else self.retrieve_num
        _, idx = self.topk(search_keys, retrieve_num)                              # :66-67
        if add_noise and self.flavour == "node":
            noise_idx = torch.randint(0, self.resource_values.shape[0], (idx.shape[0], self.noise_retrieve_num))
            idx = torch.cat([idx, noise_idx.to(idx.device)], dim=1)
        return idx

    def retrieve(self, search_keys: Tensor, search_adj, add_noise: bool, idx: Tensor | None = None):
        """ToyGraphBase.py:47-81 -> (rag_embeddings [B,k',D], rag_labels [B,k',C]).  A 1-D query (graph flavour,
        RAGraph_graph/ragraph_utils/ToyGraphBase.py:56-87) gives B = 1.  `idx`: the rows, when the caller already
        holds retrieve_indices(search_keys, add_noise) (one top-k per forward instead of two)."""
        if idx is None:
            idx = self.retrieve_indices(search_keys, add_noise)
        rag_embeddings = K.gather_rows(self.resource_values, idx)                  # :70 (+ :76,78 noise rows)
        rag_labels = K.gather_rows(self.resource_labels, idx)                      # :71 (+ :77,79)
        if add_noise and self.flavour != "node":                                   # graph :84-85,131-134
            noise = torch.normal(mean=0, std=self.noise_std, size=rag_embeddings.shape).to(rag_embeddings.device)
            rag_embeddings = K.axpby(rag_embeddings, 1.0, noise, 1.0)
        return rag_embeddings, rag_labels

    def retrieve_reduced_noisy(self, search_keys: Tensor, idx: Tensor | None = None, want_labels: bool = True):
        """What RAGraph.forward consumes in noisy fine-tuning (RAGraph.py:42-49 with add_noise): (sum_k' V, mean_k' L)
        over the top-2k rows plus the noise -- every reduction on the HIP kernels, ONE top-k per call (`idx`: the rows
        when the caller already holds retrieve_indices(search_keys, True); want_labels=False skips the label means)."""
        if idx is None:
            idx = self.retrieve_indices(search_keys, True)
        if self.flavour == "node":   # noise = extra rows: still a gather-reduce over an index matrix
            return K.gather_reduce(self.resource_values, self.resource_labels, idx)
        rag_embeddings, _ = self.retrieve(search_keys, None, True, idx=idx)   # noise is added to the gathered embeddings
        B, k, D = rag_embeddings.shape
        seg = torch.arange(0, B * k + 1, k, dtype=torch.int64, device=rag_embeddings.device)
        sum_v = K.segment_reduce(rag_embeddings.reshape(B * k, D), seg)
        mean_l = K.gather_reduce(self.resource_values, self.resource_labels, idx)[1] if want_labels else None
        return sum_v, mean_l

    def retrieve_reduced(self, search_keys: Tensor, k: int | None = None):
        """What RAGraph.forward consumes (RAGraph.py:48-49): (sum_k V[idx] [B,D], mean_k L[idx] [B,C], idx) without
        materialising the [B,k,D] gather."""
        _, idx = self.topk(search_keys, self.retrieve_num if k is None else k)
        sum_v, mean_l = K.gather_reduce(self.resource_values, self.resource_labels, idx)
        return sum_v, mean_l, idx

    # ---- persistence (the reference rebuilds its bank on every run) -------------------
    def save(self, path: str) -> None:
        """keys | values | labels as one .pt (the normalised-key cache is derived and rebuilt on load)."""
        torch.save({"format": "ragraph_amd.bank.v1", "keys": self.resource_keys.cpu(), "values": self.resource_values.cpu(),
                    "labels": self.resource_labels.cpu()}, path)

    def load(self, path: str, append: bool = False) -> None:
        blob = torch.load(path, map_location="cpu")
        if blob.get("format") != "ragraph_amd.bank.v1":
            raise ValueError(f"{path}: not a ragraph_amd bank file")
        if not append:  # fresh stores: the old ones may be tensors adopted from the caller (set_resources)
            self._keys, self._values, self._labels = (_Bank(b.buf.shape[1], self.device)
                                                      for b in (self._keys, self._values, self._labels))
        self.add_resources(blob["keys"], blob["values"], blob["labels"])

    def show(self):
        print("resource_keys", self.resource_keys.shape)
        print("resource_values", self.resource_values.shape)
        print("resource_labels", self.resource_labels.shape)
        print("resource positions", self.resource_positions.shape)
        print("label count distribution", torch.sum(self.resource_labels, dim=0))
