"""Mirror of RAGraph_node_fewshot: retrieval mixes a STRUCTURAL similarity (position-aware codes from all-pairs
shortest paths to random anchors) with the semantic cosine, the decoder is the encoder's second GCN layer, and the
label term is a prototype-logit lookup.

Scores are materialised here (B ~ 40 query nodes x a few thousand bank rows) exactly as the reference does
(ToyGraphBase.py:47-61): two cosine matrices on the HIP linear kernel, mixed with the uncontracted axpby kernel, then
the HIP top-k over rows.  Few-shot graphs are tiny; the fused streaming kernel is for the 1M-key regime.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import kernels as K
from .graph import CSRGraph, as_csr
from .ragraph_utils.Propagation import Propagation
from .ragraph_utils.ToyGraphBase import ToyGraphBase, _Bank


def _dense(adj) -> torch.Tensor:
    if isinstance(adj, torch.Tensor) and adj.layout == torch.strided:
        return adj.squeeze(0) if adj.dim() == 3 else adj
    g: CSRGraph = as_csr(adj)
    a = torch.zeros(g.n, g.n, device=g.device)
    rows = torch.repeat_interleave(torch.arange(g.n, device=g.device), g.rowptr[1:] - g.rowptr[:-1])
    a[rows, g.col.long()] = g.val
    return a


class PositionAwareEncoder:
    """ragraph_utils/PositionAwareEncoder.py: distance-to-anchor codes.  The reference runs an all-pairs Floyd-Warshall as
    n dense torch.min broadcasts and fills the code matrix in a Python double loop -- to use 10 columns of the n x n
    matrix.  The codes here come from the distances to the anchors ONLY, relaxed on the CSR of the (block-diagonal)
    batch in one launch (K.position_codes_csr: no n x n allocation, no n dependent launches); `floyd_warshall` keeps the
    reference's all-pairs entry for callers that want the matrix."""

    @staticmethod
    def floyd_warshall(adj) -> torch.Tensor:
        return K.floyd_warshall(_dense(adj))

    @staticmethod
    def encode_position_aware_code(adj, num_anchors: int, dis_q: int = 10, anchors: torch.Tensor | None = None):
        g: CSRGraph = as_csr(adj.squeeze(0) if isinstance(adj, torch.Tensor) and adj.layout == torch.strided
                             and adj.dim() == 3 else adj)
        if anchors is None:  # PositionAwareEncoder.py:11 draws them from torch's global CPU generator
            anchors = torch.randint(low=0, high=g.n, size=(int(num_anchors),))
        return K.position_codes_csr(g.rowptr, g.col, g.val, anchors.to(g.device), float(dis_q))


class ToyGraphBaseFewShot(ToyGraphBase):
    """RAGraph_node_fewshot/ragraph_utils/ToyGraphBase.py."""

    def __init__(self, pretrain_model, num_class, emb_size, query_graph_hop, retrieve_num, device="cuda"):
        super().__init__(pretrain_model, num_class, emb_size, query_graph_hop, device=device, flavour="node")
        self.retrieve_num = retrieve_num
        self.num_anchors = 10
        self.dis_q = 10
        self.structure_weight = 0.001   # :28-29
        self.semantic_weight = 0.999
        self._positions = _Bank(self.num_anchors, self.device)

    @property
    def resource_positions(self):
        return self._positions.view()

    def add_resources(self, keys, values, labels, positions=None):
        super().add_resources(keys, values, labels)
        if positions is None:
            positions = torch.zeros(keys.shape[0], self.num_anchors, device=self.device)
        self._positions.append(positions)

    def similarity_scores(self, search_keys, search_adj, anchors=None):
        """ToyGraphBase.py:49-61: structure_weight * cos(position codes) + semantic_weight * cos(embeddings)."""
        pos = PositionAwareEncoder.encode_position_aware_code(search_adj, self.num_anchors, self.dis_q, anchors)
        s_struct = K.linear(K.normalize_rows(pos), K.normalize_rows(self.resource_positions))
        s_sem = K.linear(K.normalize_rows(search_keys), self.keys_normalized)
        return K.axpby(s_struct, self.structure_weight, s_sem, self.semantic_weight)

    def retrieve(self, search_keys, search_adj, add_noise: bool, anchors=None):
        retrieve_num = 2 * self.retrieve_num if add_noise else self.retrieve_num
        scores = self.similarity_scores(search_keys, search_adj, anchors)
        _, idx = K.topk_rows(scores, retrieve_num)                                   # :64
        rag_embeddings = K.gather_rows(self.resource_values, idx)
        rag_labels = K.gather_rows(self.resource_labels, idx)
        if add_noise:                                                                # :70-76
            noise_idx = torch.randint(0, self.resource_values.shape[0], (idx.shape[0], self.noise_retrieve_num),
                                      device=idx.device)
            rag_embeddings = torch.cat([rag_embeddings, K.gather_rows(self.resource_values, noise_idx)], dim=1)
            rag_labels = torch.cat([rag_labels, K.gather_rows(self.resource_labels, noise_idx)], dim=1)
        return rag_embeddings, rag_labels


class RAGraph(nn.Module):
    """RAGraph_node_fewshot/RAGraph.py:7-83.  Forward only: the reference fine-tunes the second GCN layer through
    decode(); that backward is outside the inference path."""

    def __init__(self, pretrain_model, resource_dataset, mean_fewshot_logits, emb_size, finetune=True,
                 noise_finetune=False, query_graph_hop=3, retrieve_num=5, device="cuda", dataset_name=None):
        super().__init__()
        self.emb_size = emb_size
        self.pretrain_model = pretrain_model
        name = dataset_name or getattr(resource_dataset, "name", "ENZYMES")
        if name == "PROTEINS":                       # RAGraph.py:25-32
            self.retrieve_weight, self.label_weight = 0.3, 0.8
        else:
            self.retrieve_weight, self.label_weight = 0.5, 0.5
        self.finetune, self.noise_finetune = finetune, noise_finetune
        self.query_graph_hop = query_graph_hop
        self.toy_graph_base = ToyGraphBaseFewShot(pretrain_model, len(mean_fewshot_logits), emb_size, query_graph_hop,
                                                  retrieve_num, device=device)

    def forward(self, features, adj, mean_fewshot_logits, anchors=None):
        g = as_csr(adj)
        emb = self.pretrain_model.encode(features, g)                                           # :48
        add_noise = self.training and self.noise_finetune
        rag_embeddings, rag_labels = self.toy_graph_base.retrieve(emb, g, add_noise, anchors)    # :51
        label_ids = torch.argmax(rag_labels, dim=-1)                                             # :54 (integer lookup)
        k = label_ids.shape[1]
        rag_logits, _ = K.gather_reduce(mean_fewshot_logits, None, label_ids, v_scale=1.0 / k)   # :55,62 mean over k
        if not self.finetune:
            return rag_logits
        # :63 torch.sum(rag_embeddings, dim=1): rank-order fp32 adds over the gathered [B,k,D] rows
        flat = rag_embeddings.reshape(-1, rag_embeddings.shape[-1])
        rows = torch.arange(flat.shape[0], device=emb.device).reshape(-1, k)
        rag_embedding, _ = K.gather_reduce(flat, None, rows)
        query = Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop)               # :65
        hidden = K.axpby(query, 1 - self.retrieve_weight, rag_embedding, self.retrieve_weight)   # :67
        decode_logits = self.pretrain_model.decode(hidden, g)                                    # :69
        return K.axpby(decode_logits, 1 - self.label_weight, rag_logits, self.label_weight)      # :77


class RAGraphGraphFewShot(nn.Module):
    """RAGraph_graph_fewshot/RAGraph.py:7-91: graph classification, few-shot.  EVERY node of the one query graph is a query
    against a bank that holds every node of every resource graph with its graph's label
    (ragraph_utils/ToyGraphBase.py:118-126); the label term is a prototype-logit lookup, the decoder is the encoder's
    second GCN layer (trained through decode(): ragraph_amd.autograd.spmm_csr), and the node logits are averaged into one
    row (:86).  FewShotBase (RAGraph.py:43) only loads blobs that forward never reads and is not reproduced."""

    def __init__(self, pretrain_model, resource_dataset, feture_size, num_class, emb_size, finetune=True,
                 noise_finetune=False, device="cuda", dataset_name=None):
        super().__init__()
        self.emb_size, self.num_class, self.pretrain_model = emb_size, num_class, pretrain_model
        name = dataset_name or getattr(resource_dataset, "name", "PROTEINS")
        weights = {"ENZYMES": (0.3, 0.8), "PROTEINS": (0.5, 0.5), "COX2": (0.3, 0.6), "BZR": (0.1, 0.5)}   # RAGraph.py:16-30
        if name not in weights:
            raise NotImplementedError(name)
        self.retrieve_weight, self.label_weight = weights[name]
        self.finetune, self.noise_finetune = finetune, noise_finetune
        if noise_finetune:
            assert finetune
        self.query_graph_hop = 1                                                               # :38
        self.toy_graph_base = ToyGraphBase(pretrain_model, num_class, emb_size, self.query_graph_hop, device=device,
                                           flavour="graph_fewshot")
        if resource_dataset is not None:
            self.toy_graph_base.build_toy_graph(resource_dataset)
        self.to(device)

    def forward(self, features, adj, mean_fewshot_logits):
        from . import autograd as A
        g = as_csr(adj)
        tgb = self.toy_graph_base
        emb = self.pretrain_model.encode(features, g)                                            # :47
        add_noise = self.training and self.noise_finetune
        idx = tgb.retrieve_indices(emb, add_noise)                                               # :51 (k' = 2k with noise)
        k = idx.shape[1]
        label_ids = torch.argmax(K.gather_rows(tgb.resource_labels, idx), dim=-1)                # :55 (integer lookup)
        rag_logits, _ = K.gather_reduce(mean_fewshot_logits, None, label_ids, v_scale=1.0 / k)   # :56,68 mean over k
        if not self.finetune:
            return rag_logits                                                                    # :88-91
        if add_noise:                                                                            # graph noise: on the embeddings
            rag_embedding, _ = tgb.retrieve_reduced_noisy(emb, idx=idx, want_labels=False)
        else:
            rag_embedding, _ = K.gather_reduce(tgb.resource_values, None, idx)                   # :69 sum over k
        query = Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop)               # :71
        hidden = K.axpby(query, 1 - self.retrieve_weight, rag_embedding, self.retrieve_weight)   # :75
        decode_logits = self.pretrain_model.decode(hidden, g)                                    # :79
        label_logits = A.axpby(decode_logits, 1 - self.label_weight, rag_logits, self.label_weight)   # :82
        if label_logits.requires_grad:   # training: the mean over the nodes through torch (one reduction of [n, D])
            return label_logits.mean(dim=0).unsqueeze(0)                                         # :84
        seg = torch.tensor([0, label_logits.shape[0]], dtype=torch.int64, device=label_logits.device)
        return K.segment_reduce(label_logits, seg, mean_mode=True)                               # :84
