"""Mirror of RAGraph_edge/modules/RAGraph.py: LightGCN-style propagation with time-softmax edge weights, retrieval
over all users+items, fusion -- and its fine-tuning step (cal_loss: edge dropout, the forward with gradients through
gate / LoRA factors / three propagation layers, BPR + L2; modules/RAGraph.py:121-171,335-376), every product of forward
AND backward on the HIP kernels (ragraph_amd.autograd).  Retrieval is not differentiated (the bank carries no gradient
and torch.topk's indices have none); the trainers, metrics and loggers around cal_loss stay out of scope.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import autograd as A
from . import kernels as K
from .graph import CSRGraph


class RAGraph(nn.Module):
    def __init__(self, dataset, pretrained_model=None, phase="finetune", use_RAG=True, use_noise=False,
                 use_LoRA=False, LoRA_rank=16, emb_size=64, num_layers=3, retrieve_num=10, retrieve_weight=0.3,
                 batch_size=4096, device="cuda", num_augment_scale=0, num_inverse_sample=0):
        """dataset: .num_users, .num_items, .edges [2E,2] int64 (src,dst, both directions), .edge_norm [2E] fp32,
        .edge_times [2E] int64 (the tensors modules/RAGraph.py:22-27 derives from the scipy graph).
        pretrained_model: .generate() -> (user_emb, item_emb)."""
        super().__init__()
        self.use_LoRA = bool(use_LoRA) and phase == "finetune"
        self.edge_dropout, self.weight_decay, self.emb_dropout = 0.5, 1e-4, 0.0   # utils/parse_args.py:22,27,35 defaults
        self.num_users, self.num_items = dataset.num_users, dataset.num_items
        self.emb_size, self.num_layers = emb_size, num_layers
        self.edges = dataset.edges.to(device)
        self.edge_norm = dataset.edge_norm.to(device).float()
        self.edge_times = dataset.edge_times.to(device)
        self.phase, self.use_RAG = phase, use_RAG
        self.use_noise = use_noise and phase == "finetune"
        self.retrieve_weight, self.retrieve_num, self.batch_size = retrieve_weight, retrieve_num, batch_size  # :33-85
        self.noise_retrieve_num = 1
        # bank construction (modules/RAGraph.py:38-44,56-62): the vanilla phase keeps an inverse-importance SAMPLE of the
        # nodes (round(0.01 n) draws) of the original and of num_augment_scale feature-augmented copies
        self.num_augment_scale, self.num_inverse_sample = num_augment_scale, num_inverse_sample
        self.resource_keys = self.resource_values = None
        self._keys_normalized = self._index = None
        self._csr_cache = None
        ue, ie = pretrained_model.generate()
        self.user_embedding = nn.Parameter(ue.detach().clone().to(device))
        self.item_embedding = nn.Parameter(ie.detach().clone().to(device))
        if phase == "finetune":   # :163-168
            self.gating_weight = nn.Parameter(nn.init.xavier_uniform_(torch.empty(emb_size, emb_size, device=device)))
            self.gating_bias = nn.Parameter(nn.init.xavier_uniform_(torch.empty(1, emb_size, device=device)))
        else:
            self.gating_weight = self.gating_bias = None
        if self.use_LoRA:   # :121-160: rank-r factors of the pretrained tables (U_r S_r, V_r^T), trained next to them
            for name, emb in (("user", self.user_embedding), ("item", self.item_embedding)):
                U, S, V = torch.svd(emb.detach())
                setattr(self, f"{name}_embedding_A", nn.Parameter((U[:, :LoRA_rank] @ torch.diag(S[:LoRA_rank])).contiguous()))
                setattr(self, f"{name}_embedding_B", nn.Parameter(V[:, :LoRA_rank].t().contiguous()))
        if use_RAG:
            self._make_resource_graph(pretrained_model)

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _csr(self, edges):
        """Destination-sorted CSR of an edge list (stable: keeps scatter_add_'s accumulation order), cached per edge
        tensor: the cache holds the tensor itself and its version counter, so another list of the same length at a
        recycled address, or an in-place edit, rebuilds it."""
        c = self._csr_cache
        if c is None or c[0] is not edges or c[1] != edges._version:
            n = self.num_users + self.num_items
            g, perm = CSRGraph.from_coo(edges[:, 1], edges[:, 0], torch.ones(edges.shape[0], device=edges.device), n)
            self._csr_cache = c = (edges, edges._version, g, perm)
        return c[2], c[3]

    def _gate_wt(self):
        """gating_weight^T (the linear kernel takes nn.Linear layout), re-made only when the parameter changes."""
        w = self.gating_weight
        tag = (w.data_ptr(), w._version)
        if getattr(self, "_gate_cache", (None, None))[0] != tag:
            self._gate_cache = (tag, w.detach().t().contiguous())
        return self._gate_cache[1]

    def emb_gate(self, x):
        """modules/RAGraph.py:168: x * sigmoid(x @ W + b) (emb_dropout p = 0 by default)."""
        if self.gating_weight is None:
            return x
        if torch.is_grad_enabled() and (x.requires_grad or self.gating_weight.requires_grad):
            z = A.linear(x, self.gating_weight.t().contiguous(), self.gating_bias.reshape(-1))
            out = A.sigmoid_gate(x, z)
            return torch.nn.functional.dropout(out, self.emb_dropout, self.training) if self.emb_dropout > 0 else out
        z = K.linear(x, self._gate_wt(), self.gating_bias.reshape(-1))
        return K.sigmoid_gate(x, z)

    def _embeddings(self):
        """:269-275: the tables, plus the LoRA product A @ B when fine-tuning with LoRA."""
        ue, ie = self.user_embedding, self.item_embedding
        if self.use_LoRA:
            ue = ue + A.linear(self.user_embedding_A, self.user_embedding_B.t().contiguous())
            ie = ie + A.linear(self.item_embedding_A, self.item_embedding_B.t().contiguous())
        return ue, ie

    def _time_range(self, edge_times, max_time_step):
        """(min, max) of the time steps as host scalars (kernel arguments), cached per edge-time tensor (identity and
        version, as _csr)."""
        c = getattr(self, "_trange", None)
        if c is None or c[0] is not edge_times or c[1] != edge_times._version:
            self._trange = c = (edge_times, edge_times._version, float(edge_times.min()), float(edge_times.max()))
        return c[2], (c[3] if max_time_step is None else float(max_time_step))

    def _agg(self, all_emb, edges, edge_norm):
        """modules/RAGraph.py:232-240: out[dst] += emb[src] * norm, as one CSR SpMM (no atomics)."""
        g, perm = self._csr(edges)
        return K.spmm_csr(g.rowptr, g.col, edge_norm[perm].contiguous(), all_emb, long_rows=g.has_long_rows)

    def _relative_edge_time_encoding(self, edges, edge_times, max_step=None):
        """modules/RAGraph.py:250-263.  Returns the softmax in ORIGINAL edge order."""
        g, perm = self._csr(edges)
        tmin, tmax = self._time_range(edge_times, max_step)
        t = K.time_rescale(edge_times, tmin, tmax)
        sm = K.segment_softmax(g.rowptr, t[perm].contiguous(), long_rows=g.has_long_rows)
        out = torch.empty_like(sm)
        out[perm] = sm
        return out

    query_shard = None  # ragraph_amd.sharded.QueryShard: retrieve only this rank's rows of the node set (inference)

    def _make_resource_graph(self, pretrained_model):
        """modules/RAGraph.py:185-226 with num_augment_scale = num_inverse_sample = 0 (the finetune-phase settings,
        :45-50): keys = embeddings after num_layers aggregations, values = sum of the even layers."""
        ue, ie = pretrained_model.generate()
        all_emb = torch.cat([ue, ie], dim=0).to(self.edges.device).float()
        res = [all_emb]
        for _ in range(self.num_layers):
            res.append(self._agg(res[-1], self.edges, self.edge_norm))
        vals = res[0]
        for r in res[2::2]:
            vals = K.axpby(vals, 1.0, r, 1.0)
        keys = res[-1]
        if self.num_inverse_sample > 0 or self.num_augment_scale > 0:
            keys, vals = self._sample_bank(keys, vals)
        self.resource_keys, self.resource_values = keys, vals
        self._keys_normalized = self._index = None

    def sample_prob(self):
        """InverseSampling.compute_sample_prob(self.adj) (modules/ragraph_utils/InverseSampling.py:6-19) on the
        bi-normalised bipartite adjacency: sparse PageRank + degree centrality on the HIP kernels, no host round trip."""
        from .bank_build import compute_sample_prob
        n = self.num_users + self.num_items
        g, _ = CSRGraph.from_coo(self.edges[:, 0], self.edges[:, 1], self.edge_norm, n, sort_cols=True)
        return compute_sample_prob(g)

    def _sample_bank(self, all_emb, all_logits):
        """modules/RAGraph.py:199-226: the original embeddings and num_augment_scale noisy / dropped copies
        (Augmentation.augment_features, modules/ragraph_utils/Augmentation.py:8-22), each cut down to num_inverse_sample
        rows drawn with replacement by inverse importance.  The draws come from torch's RNG (as the reference's)."""
        prob = self.sample_prob()                                                  # :201
        keys_out, vals_out = [], []
        for i in range(1 + self.num_augment_scale):                                # :203-204
            k, v = all_emb, all_logits
            if i > 0:                                                              # :206-208
                def aug(x):
                    noisy = x + torch.randn_like(x) * 0.1
                    return noisy * torch.bernoulli(prob * 0.01).unsqueeze(-1)
                k, v = aug(all_emb), aug(all_logits)
            if self.num_inverse_sample > 0:                                        # :214-218
                mask = torch.multinomial(prob, num_samples=self.num_inverse_sample, replacement=True)
                k, v = K.gather_rows(k, mask), K.gather_rows(v, mask)
            keys_out.append(k)
            vals_out.append(v)
        return torch.cat(keys_out, 0), torch.cat(vals_out, 0)                      # :220-226

    @property
    def keys_normalized(self):
        if self._keys_normalized is None:
            self._keys_normalized = K.normalize_rows(self.resource_keys)
        return self._keys_normalized

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward(self, edges, edge_norm, edge_times, max_time_step=None):
        """modules/RAGraph.py:265-333."""
        g, perm = self._csr(edges)
        tmin, tmax = self._time_range(edge_times, max_time_step)
        t = K.time_rescale(edge_times[perm].contiguous(), tmin, tmax)                           # :254-257
        time_norm = K.segment_softmax(g.rowptr, t, long_rows=g.has_long_rows)                  # :266
        norm = K.axpby(edge_norm[perm].contiguous(), 0.5, time_norm, 0.5)                      # :267
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        ue, ie = self._embeddings()
        all_emb = self.emb_gate(torch.cat([ue, ie], dim=0))                                     # :276-277
        if not train:
            all_emb = all_emb.detach()
        res = [all_emb]
        gw = CSRGraph(g.rowptr, g.col, norm, g.n) if train else None   # (this step's weights; its transpose serves backward)
        for _ in range(self.num_layers):                                                       # :280-283
            res.append(A.spmm_csr(gw, res[-1]) if train else
                       K.spmm_csr(g.rowptr, g.col, norm, res[-1], long_rows=g.has_long_rows))
        total = res[0]
        for r in res[1:]:                                                                      # :327 sum(res_emb)
            total = A.axpby(total, 1.0, r, 1.0)
        if self.use_RAG and self.phase in ("vanilla", "finetune"):
            add_noise = self.use_noise and self.training
            k = self.retrieve_num + (self.noise_retrieve_num if add_noise else 0)             # :308
            # :298-324: the reference walks the queries in slabs of batch_size only to bound its B x N score matrix;
            # the fused kernel never builds that matrix, so all queries go in one launch.
            if self._index is None:
                self._index = K.KeyIndex(self.keys_normalized)
            # Several GPUs (c5): the bank is replicated and the queries -- independent of one another -- are split;
            # a rank retrieves and reduces its rows, one all_gather of the [n, D] means completes the step
            # (ragraph_amd.sharded.QueryShard; the result does not depend on the split, bit for bit).
            qs = self.query_shard if not self.training else None
            queries = res[0].detach()
            if qs is not None:
                lo, hi = qs.bounds(res[0].shape[0])
                queries = res[0][lo:hi].contiguous()
            if k > K.N.TOPK_MAX:
                # vanilla phase (:57,73: retrieve_num = 50 ... 100000): only the winners' MEAN is consumed (:321), so the
                # top-k SET is selected from score slabs (radix select) instead of sorted lists
                rag = K.retrieve_mean_large_k(queries, self.keys_normalized, self.resource_values, k)
                if qs is not None:
                    rag = qs.gather_rows(rag, res[0].shape[0])
                total = A.axpby(total, 1 - self.retrieve_weight, rag, self.retrieve_weight)    # :328
                return total.split([self.num_users, self.num_items], dim=0)
            _, idx = self._index.topk(queries, k)
            if add_noise:
                # (drawn from the default CPU generator as the reference does, :316; its per-slab draws of
                # [batch, 1] concatenate to this one [n, 1] draw, so torch.manual_seed reproduces its rows)
                noise = torch.randint(0, self.resource_values.shape[0], (idx.shape[0], self.noise_retrieve_num)
                                      ).to(idx.device)
                idx = torch.cat([idx, noise], dim=1)
            rag, _ = K.gather_reduce(self.resource_values, None, idx, v_scale=1.0 / idx.shape[1])  # :314,321 mean
            if qs is not None:
                rag = qs.gather_rows(rag, res[0].shape[0])
            total = A.axpby(total, 1 - self.retrieve_weight, rag, self.retrieve_weight)        # :328
        return total.split([self.num_users, self.num_items], dim=0)

    @torch.no_grad()
    def generate(self, max_time_step=None):
        return self.forward(self.edges, self.edge_norm, self.edge_times, max_time_step=max_time_step)

    @torch.no_grad()
    def rating(self, user_emb, item_emb):
        return K.linear(user_emb, item_emb)   # modules/RAGraph.py:362-364: user_emb @ item_emb.T

    # ---- fine-tuning step --------------------------------------------------------------------------------------------
    dropout_rng = "host"   # "host": the reference's draw (torch.rand on the CPU generator, utils.py:46) -- the same mask as the
                           # reference for the same seed, at the cost of one uniform per edge on the host and a copy per step
                           # (44 M edges: ~0.3 s of a 1-s step at c5, bench `finetune_step.edge_c5.host_mask_draw_ms`);
                           # "device": torch.rand on the device generator (another stream of random numbers, same law)

    def draw_edge_mask(self):
        """The step's edge-dropout mask (bool, on the device), or None when nothing is dropped."""
        keep = 1.0 - self.edge_dropout
        if keep >= 1.0:
            return None
        n_e = self.edges.shape[0]
        if self.dropout_rng == "device":
            return (torch.rand(n_e, device=self.edges.device) + keep).floor().bool()
        return (torch.rand(n_e) + keep).floor().bool().to(self.edges.device)

    def cal_loss(self, batch_data):
        """modules/RAGraph.py:335-355: edge dropout (keep 1 - edge_dropout; the reference draws the mask with torch.rand on
        the CPU generator, utils.py:46), forward on the kept edges, BPR loss on (user, positive, negative) triples + L2 on
        the batch's table rows.  Returns (loss, {"rec_loss", "reg_loss"})."""
        mask = self.draw_edge_mask()
        if mask is None:
            edges, norm, times = self.edges, self.edge_norm, self.edge_times
        else:
            kept = K.mask_positions(mask)            # (the library's own prefix sums: no other library's select on this path)
            edges, norm, times = self.edges[kept], self.edge_norm[kept], self.edge_times[kept]
        users, pos_items, neg_items = (t.to(self.edges.device).long() for t in batch_data)
        user_emb, item_emb = self.forward(edges, norm, times)
        u = A.gather_rows(user_emb.contiguous(), users)                                        # :343-345
        p = A.gather_rows(item_emb.contiguous(), pos_items)
        q = A.gather_rows(item_emb.contiguous(), neg_items)
        # base_model.py:81-86 (the loss itself: a few thousand scalars -- trainer-side bookkeeping in torch)
        pos_score, neg_score = (u * p).sum(dim=1), (u * q).sum(dim=1)
        rec = (-torch.log(1e-10 + torch.sigmoid(pos_score - neg_score))).mean()
        ue, ie = self._embeddings()                                                            # :365-376
        ru, rp, rq = A.gather_rows(ue, users), A.gather_rows(ie, pos_items), A.gather_rows(ie, neg_items)
        reg = 0.5 * (ru.norm(2).pow(2) + rp.norm(2).pow(2) + rq.norm(2).pow(2)) / float(len(users))
        loss = rec + self.weight_decay * reg
        return loss, {"rec_loss": float(rec), "reg_loss": float(self.weight_decay * reg)}
