"""Mirror of RAGraph_node/RAGraph.py and RAGraph_graph/RAGraph.py: encode -> retrieve -> propagate -> fuse -> decode.

Same constructor and forward signatures as the reference (including its `feture_size` spelling); hyper-parameters are
the reference's hard-coded constants exposed as attributes.
"""
import os

import torch
import torch.nn as nn

from . import autograd as A
from . import kernels as K
from .graph import as_csr
from .ragraph_utils import Propagation, TaskDecoder, ToyGraphBase


class RAGraph(nn.Module):
    """Node classification flavour -- RAGraph_node/RAGraph.py:10-63."""

    flavour = "node"

    def __init__(self, pretrain_model, resource_dataset, feture_size, num_class, emb_size, finetune=True,
                 noise_finetune=False, device="cuda") -> None:
        super().__init__()
        self.emb_size = emb_size
        self.num_class = num_class
        self.pretrain_model = pretrain_model
        self.retrieve_weight, self.label_weight, self.query_graph_hop = self._hyper()
        self.finetune = finetune
        self.noise_finetune = noise_finetune
        if self.noise_finetune:
            assert self.finetune
        self.toy_graph_base = ToyGraphBase(pretrain_model, num_class, emb_size, self.query_graph_hop, device=device,
                                           flavour=self.flavour)
        if resource_dataset is not None:
            self.toy_graph_base.build_toy_graph(resource_dataset)
        if self.finetune:
            self.decoder = TaskDecoder(emb_size, emb_size, num_class)
            self.reset_parameters()
        self.to(device)

    def _hyper(self):
        return 0.5, 0.5, 3  # RAGraph_node/RAGraph.py:18-19,26

    def reset_parameters(self):
        self.decoder.reset_parameters()

    def _queries(self, emb, g):
        return emb

    def _pool(self, x, g):
        return x

    query_shard = None  # ragraph_amd.sharded.QueryShard: answer only this rank's rows of the batch (inference)

    def forward(self, features, adj):
        g = as_csr(adj)
        tgb = self.toy_graph_base
        if self.query_shard is not None and not self.training and self.flavour == "node":
            hybrid = (getattr(tgb, "values_replicated", False) and hasattr(tgb, "retrieve_reduced_rows")
                      and (tgb.collective or tgb.emulate_world > 1))
            sliced = self._forward_sliced_encode(features, g, hybrid)
            if sliced is not None:
                return sliced
        pretrain_embedddings = self.pretrain_model.inference(features, g)                      # RAGraph.py:40
        add_noise = self.training and self.noise_finetune
        queries = self._queries(pretrain_embedddings, g)
        if self.query_shard is not None and not self.training and self.flavour == "node":
            if (getattr(tgb, "values_replicated", False) and hasattr(tgb, "retrieve_reduced_rows")
                    and (tgb.collective or tgb.emulate_world > 1)):
                return self._forward_hybrid(queries, pretrain_embedddings, g)
            return self._forward_query_shard(queries, pretrain_embedddings, g)
        if (not self.training and self.flavour == "node" and getattr(tgb, "values_replicated", False)
                and hasattr(tgb, "retrieve_reduced_rows") and (tgb.collective or tgb.emulate_world > 1)):
            return self._forward_key_shard(queries, pretrain_embedddings, g)
        # The k-hop propagation needs only the embeddings, not the retrieval: on a large graph (inference) it is enqueued on
        # a side stream, where its SpMM launches fill the latency-bound rescoring phases between the retrieval's matrix
        # passes instead of running behind them (c2: 0.4 ms of 37.5).  Same kernels, same bits; one fork / join.
        side = None
        # (training too: the encoder's output is detached -- preprompt.py:62 -- so the hops carry no tape; only a caller that
        # does ask gradients of the embeddings keeps them on the main stream, inside autograd's stream bookkeeping)
        if (self.finetune and not (torch.is_grad_enabled() and pretrain_embedddings.requires_grad) and queries.is_cuda
                and queries.shape[0] >= self.OVERLAP_MIN_NODES):
            # (small forwards gain nothing from the fork, captured in a HIP graph or not: Cora-sized replay 0.163 ms
            # without, 0.172 with the hops on a parallel branch -- round 3)
            main = torch.cuda.current_stream()
            side = self._side_stream(queries.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                query_embeddings = self._pool(
                    Propagation.aggregate_k_hop_features(g, pretrain_embedddings, self.query_graph_hop), g)
        if (side is not None and not add_noise and not torch.is_grad_enabled() and self.flavour == "node"
                and type(tgb).retrieve_reduced is ToyGraphBase.retrieve_reduced
                and os.environ.get("RAGRAPH_GATHER_MIX", "1") != "0"):   # (=0: the separate entries, A/B)
            # a large inference forward: the winners' value sum goes straight into the prompt fusion (:48-49 + :53 in one
            # launch -- the [n, D] sum is never written, the axpby launch is gone); same bits as the separate entries below
            _, idx = tgb.topk(queries, tgb.retrieve_num)                                       # :43
            main.wait_stream(side)
            query_embeddings.record_stream(main)
            hidden, rag_label = K.gather_reduce_mix(tgb.resource_values, tgb.resource_labels, idx, query_embeddings,
                                                    1 - self.retrieve_weight, self.retrieve_weight)
            return A.softmax_mix(self.decoder(hidden), rag_label, self.label_weight)           # :54-57
        if add_noise:
            rag_embedding, rag_label = tgb.retrieve_reduced_noisy(queries)                     # :43,48-49 (noise branch)
        else:
            rag_embedding, rag_label, _ = tgb.retrieve_reduced(queries)                        # :43,48-49 fused
        if not self.finetune:
            return rag_label                                                                   # :60-63
        if side is not None:
            main.wait_stream(side)
            query_embeddings.record_stream(main)
        else:
            query_embeddings = self._pool(
                Propagation.aggregate_k_hop_features(g, pretrain_embedddings, self.query_graph_hop), g)  # :51
        return self._fuse_decode(query_embeddings, rag_embedding, rag_label)                   # :53-57

    OVERLAP_MIN_NODES = 16384   # below: the fork / join costs more than the overlap gives

    def _side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            self._side = st
        return st

    def _fuse_decode(self, query_embeddings, rag_embedding, rag_label):
        """RAGraph.py:53-57: prompt fusion -> task decoder -> label mix.  An inference forward of a small batch (Cora,
        PROTEINS: launch-bound) takes the single fused launch; training and large batches the separate differentiable
        entries (MFMA fc1).  Same bits either way."""
        fc1, fc2 = self.decoder.layers()
        trains = torch.is_grad_enabled() and (query_embeddings.requires_grad or rag_embedding.requires_grad or
                                              any(p.requires_grad for p in self.decoder.parameters()))
        if not trains and K.fuse_decode_fits(query_embeddings.shape[0], fc1.in_features, fc1.out_features, fc2.out_features):
            return K.fuse_decode(query_embeddings, rag_embedding, 1 - self.retrieve_weight, self.retrieve_weight,
                                 fc1.weight, fc1.bias, self.decoder.NEGATIVE_SLOPE, fc2.weight, fc2.bias, rag_label,
                                 self.label_weight)
        hidden = A.axpby(query_embeddings, 1 - self.retrieve_weight, rag_embedding, self.retrieve_weight)  # :53
        decode_label = self.decoder(hidden)                                                    # :54
        return A.softmax_mix(decode_label, rag_label, self.label_weight)                       # :55-57


    def _forward_sliced_encode(self, features, g, hybrid: bool):
        """Query-sharded ranks on a large graph (round 6): a rank's retrieval needs the embeddings of ITS slice of the queries
        only, so the main stream encodes those rows alone (PrePrompt.inference_rows: the last layer over a slice of the row
        pointers -- 0.02 ms instead of 0.17 at c2 for an eighth of the nodes) and starts the retrieval at once; the whole-graph
        encode that the k-hop propagation needs (every node's embedding feeds somebody's hops) runs on the side stream with the
        hops, under the retrieval.  Row for row the arithmetic of forward().  Emulated rank of 8: 3.28 -> 3.12 ms per step.
        Not for the hybrid layout (measured 3.06 -> 3.25: its retrieval is a chain of short launches with exchanges between
        them, and the dense encode on the side stream holds LDS the filter kernel's workgroups wait for).  None: plain path."""
        x = features.squeeze(0) if features.dim() == 3 else features
        n = x.shape[0]
        if hybrid or not (self.finetune and x.is_cuda and n >= self.OVERLAP_MIN_NODES):
            return None
        qs, tgb = self.query_shard, self.toy_graph_base
        qlo, qhi = qs.bounds(n)
        q = self.pretrain_model.inference_rows(features, g, qlo, qhi)   # my queries, main stream
        if q is None:
            return None
        main = torch.cuda.current_stream()
        side = self._side_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            emb = self.pretrain_model.inference(features, g)
            query_embeddings = Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop, rows=(qlo, qhi))
        rag_embedding, rag_label, _ = tgb.retrieve_reduced(q)
        main.wait_stream(side)
        query_embeddings.record_stream(main)
        return qs.gather_rows(self._fuse_decode(query_embeddings, rag_embedding, rag_label), n)

    def _forward_query_shard(self, queries, emb, g):
        """Multi-GPU inference with a replicated bank: the cheap graph part runs on every rank, the retrieval and the
        decoder only on this rank's slice of the nodes; one all_gather of the [n, C] outputs puts the whole result on
        every rank.  Row for row the same arithmetic as forward()."""
        qs = self.query_shard
        n = queries.shape[0]
        lo, hi = qs.bounds(n)
        rag_embedding, rag_label, _ = self.toy_graph_base.retrieve_reduced(queries[lo:hi].contiguous())
        if not self.finetune:
            return qs.gather_rows(rag_label, n)
        query_embeddings = Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop)[lo:hi].contiguous()
        return qs.gather_rows(self._fuse_decode(query_embeddings, rag_embedding, rag_label), n)


    def _forward_key_shard(self, queries, emb, g):
        """Multi-GPU inference with a row-sharded KEY bank (ragraph_amd.sharded.ShardedToyGraphBase, values replicated):
        every rank filters ALL queries against its shard; everything behind that is per query and runs on the rank's
        slice of the nodes only -- merge of the per-shard lists (one all_to_all), value / label gather, the last
        propagation hop, fusion + decoder -- and one all_gather of the [n, C] outputs completes the step.  Row for row
        the arithmetic of forward()."""
        tgb = self.toy_graph_base
        n = queries.shape[0]
        lo, hi = tgb.tail_bounds(n)
        side = None
        if self.finetune and queries.is_cuda and n >= self.OVERLAP_MIN_NODES:  # (see forward(): propagation on a side stream)
            main = torch.cuda.current_stream()
            side = self._side_stream(queries.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                query_embeddings = Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop, rows=(lo, hi))
        # (the filtered call may run under the group's speculative first bound: its proof -- the merged k-th best of every row
        # against the prior, pooled over the ranks -- is read AFTER the rest of the tail has been enqueued, just before the
        # output leaves; a miss anywhere repeats the retrieval without the prior on every rank alike)
        rag_embedding, rag_label, _ = tgb.retrieve_reduced_rows(queries, defer_verify=True)
        if side is not None:
            main.wait_stream(side)
            query_embeddings.record_stream(main)
        elif self.finetune:
            query_embeddings = Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop, rows=(lo, hi))
        out = self._fuse_decode(query_embeddings, rag_embedding, rag_label) if self.finetune else rag_label
        if not tgb.confirm():
            rag_embedding, rag_label, _ = tgb.retrieve_reduced_rows(queries)
            out = self._fuse_decode(query_embeddings, rag_embedding, rag_label) if self.finetune else rag_label
        return tgb.gather_output_rows(out, n)


    def _forward_hybrid(self, queries, emb, g):
        """Multi-GPU inference on Q query groups x S key shards (ragraph_amd.sharded.HybridLayout): `query_shard` splits the
        batch over the query groups, `toy_graph_base` is the row-sharded bank of this rank's key group.  A rank filters its
        group's slice of the queries against its key shard (exchanges with the S - 1 partners of its group only), finishes
        its share of those rows -- merge, gathers, last hop, decoder -- and two all_gathers of [., C] outputs (key group,
        query axis) put the whole result on every rank.  Row for row the arithmetic of forward()."""
        from .sharded import HybridLayout as H

        qs, tgb = self.query_shard, self.toy_graph_base
        n = queries.shape[0]
        qlo, qhi, lo, hi = H.rows(qs, tgb, n)
        rows = (qlo + lo, qlo + hi)
        side = None
        if self.finetune and queries.is_cuda and n >= self.OVERLAP_MIN_NODES:  # (see forward(): propagation on a side stream)
            main = torch.cuda.current_stream()
            side = self._side_stream(queries.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                query_embeddings = Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop, rows=rows)
        # (speculative first bound: decided, proven and withdrawn inside the KEY group -- see _forward_key_shard; the verdict is
        # read before any collective of the query axis, so the query groups never disagree on how many of those run)
        rag_embedding, rag_label, _ = H.retrieve_reduced_rows(qs, tgb, queries, defer_verify=True)
        if side is not None:
            main.wait_stream(side)
            query_embeddings.record_stream(main)
        elif self.finetune:
            query_embeddings = Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop, rows=rows)
        out = self._fuse_decode(query_embeddings, rag_embedding, rag_label) if self.finetune else rag_label
        if not tgb.confirm():
            rag_embedding, rag_label, _ = H.retrieve_reduced_rows(qs, tgb, queries)
            out = self._fuse_decode(query_embeddings, rag_embedding, rag_label) if self.finetune else rag_label
        return H.gather_output_rows(qs, tgb, out, n)


class RAGraphGraph(RAGraph):
    """Graph classification flavour -- RAGraph_graph/RAGraph.py:7-75: one mean-pooled query per graph."""

    flavour = "graph"

    def _hyper(self):
        return 0.3, 0.3, 1  # RAGraph_graph/RAGraph.py:25-26,33

    @staticmethod
    def _mean_rows(x):
        seg = torch.tensor([0, x.shape[0]], dtype=torch.int64, device=x.device)
        return K.segment_reduce(x, seg, mean_mode=True)  # [1,D]

    def _queries(self, emb, g):
        return self._mean_rows(emb)        # RAGraph_graph/RAGraph.py:50 (1-D query -> one row)

    def _pool(self, x, g):
        return self._mean_rows(x)          # :63

    @torch.no_grad()
    def forward_batch(self, features, adj, graph_ptr):
        """G graphs in one pass (the reference runs batch_size = 1, finetune-rag.py:27): `adj` is the block-diagonal CSR
        of the batch, `graph_ptr` [G+1] the node offsets.  Row g of the result equals forward() on graph g alone --
        every step is per-node or per-segment, and a score does not depend on the query batch."""
        g = as_csr(adj)
        seg = graph_ptr.to(features.device, torch.int64)
        emb = self.pretrain_model.inference(features, g)                                        # :49
        rag_embedding, rag_label, _ = self.toy_graph_base.retrieve_reduced(
            K.segment_reduce(emb, seg, mean_mode=True))                                          # :50-60, G queries
        if not self.finetune:
            return rag_label
        query = K.segment_reduce(Propagation.aggregate_k_hop_features(g, emb, self.query_graph_hop), seg,
                                 mean_mode=True)                                                 # :62-63
        return self._fuse_decode(query, rag_embedding, rag_label)                                # :65-69
