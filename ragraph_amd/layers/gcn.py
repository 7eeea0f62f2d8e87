"""Mirror of RAGraph_*/layers/gcn.py: one GCN layer PReLU(A_hat (X W^T) + b), same parameter names (fc, act, bias)."""
import os
import weakref

import torch
import torch.nn as nn

from .. import kernels as K
from ..graph import as_csr

# Bag-of-words node features (Cora / Citeseer / Pubmed: 1-2 % non-zeros over 500 - 3700 columns) make X W^T a SPARSE
# product: the dense kernel's chain fmaf(x_k, w_k, acc) over k = 0..F-1 from +0 leaves acc unchanged wherever x_k = 0
# (acc + (+-0) = acc, and +0 + (-0) = +0: an accumulator that starts at +0 never becomes -0), so the CSR SpMM of X's
# non-zeros against W^T -- the same fmaf chain in column order -- gives the SAME BITS for finite weights (a non-finite
# weight would turn the skipped 0 * w into NaN in the dense chain) at 1/50 of the work: 2708 x 1433 -> 128: 35 -> 6 us.
# The feature matrix is the dataset's and the same tensor every forward: its CSR form is made once per tensor version.
SPARSE_FEATURES_MAX_DENSITY = 0.05
SPARSE_FEATURES_MIN_COLS = 256
_feature_csr = {}  # id(base tensor) -> (weakref to it, (version, data_ptr, shape, stride) of the view, (rowptr, col, val) or None)


def sparse_features(x: torch.Tensor, probe: bool = True):
    """(rowptr, col, val) of a feature matrix worth multiplying as a sparse one, else None.  Decided once per tensor
    version (one count + one synchronisation; never while a HIP graph is being captured: an unjudged tensor is dense).
    probe=False only looks the tensor up: the encoder's entry points (PrePrompt) judge what they are GIVEN -- the
    dataset's features -- and the layers never probe their inputs, most of which are activations made anew (and at the
    same address) every forward.  The entry follows the tensor's version counter: features changed in place are judged
    again, except through `.data` (which has a counter of its own) -- make a new tensor for new features."""
    if x.dim() == 3 and x.shape[0] == 1:
        x = x[0]
    if x.dim() != 2 or x.shape[1] < SPARSE_FEATURES_MIN_COLS or x.shape[1] > K.ROW_BLOCK or x.numel() == 0 or not x.is_cuda:
        return None
    # (the encoder hands every layer a fresh VIEW of the caller's tensor -- torch.squeeze -- so the entry hangs on the
    # view's base, which is the caller's object, and remembers which view of it was judged)
    base = x._base if x._base is not None else x
    sig = (x._version, x.data_ptr(), tuple(x.shape), tuple(x.stride()))
    ent = _feature_csr.get(id(base))
    if ent is not None and ent[0]() is base and ent[1] == sig:
        return ent[2]
    if not probe or torch.cuda.is_current_stream_capturing():
        return None
    csr = None
    if int(torch.count_nonzero(x)) <= SPARSE_FEATURES_MAX_DENSITY * x.numel():
        xs = x.detach().to_sparse_csr()
        csr = (xs.crow_indices().contiguous(), xs.col_indices().to(torch.int32).contiguous(), xs.values().contiguous())
    if len(_feature_csr) > 64:  # (entries of tensors that are gone)
        for key in [k_ for k_, v_ in _feature_csr.items() if v_[0]() is None]:
            del _feature_csr[key]
    _feature_csr[id(base)] = (weakref.ref(base), sig, csr)
    return csr


def aggregate_first(f_in: int, f_out: int) -> bool:
    """Inference of a layer whose input is at most half as wide as its output (and a width the SpMM takes): aggregate the
    narrow features, then multiply.  (Training keeps the reference's order: autograd.linear -> autograd.spmm_csr.)
    Another association of the same sum: the embeddings differ from the reference order's by fp32 rounding (<= 1e-5), which
    can flip a near-tie of the top-k; RAGRAPH_GCN_REFERENCE_ORDER=1 (read per call) keeps A_hat (X W^T) everywhere --
    inference then has the training path's bits."""
    if os.environ.get("RAGRAPH_GCN_REFERENCE_ORDER") == "1":
        return False
    return f_in % 4 == 0 and f_in >= 16 and 2 * f_in <= f_out


class GCN(nn.Module):
    def __init__(self, in_ft, out_ft, act=None, bias=True):
        super().__init__()
        self.fc = nn.Linear(in_ft, out_ft, bias=False)  # parameter container; math runs in libragraph_hip
        self.act = nn.PReLU()
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_ft))
        else:
            self.register_parameter("bias", None)
        torch.nn.init.xavier_uniform_(self.fc.weight.data)  # layers/gcn.py:18-24
        self._alpha_cache = (None, 0.25)
        self._wt_cache = (None, None)
        # (the caches follow the parameters' version counters; a checkpoint load starts them over -- writes through `.data`
        # bypass the counters: assign parameters with copy_() / load_state_dict)
        self._register_load_state_dict_pre_hook(lambda *a, **kw: self._drop_caches())

    def _drop_caches(self):
        self._alpha_cache = (None, 0.25)
        self._wt_cache = (None, None)

    def _alpha(self) -> float:
        """PReLU slope as a host scalar (a kernel argument), re-read only when the parameter changes."""
        w = self.act.weight
        tag = (w.data_ptr(), w._version)
        if self._alpha_cache[0] != tag:
            self._alpha_cache = (tag, float(w.detach().reshape(-1)[0]))
        return self._alpha_cache[1]

    def _weight_t(self) -> torch.Tensor:
        """fc.weight^T [F, out] (the SpMM's dense operand), re-made only when the parameter changes."""
        w = self.fc.weight
        tag = (w.data_ptr(), w._version)
        if self._wt_cache[0] != tag:
            self._wt_cache = (tag, w.detach().t().contiguous())
        return self._wt_cache[1]

    def forward_rows(self, x, adj, lo: int, hi: int):
        """Rows [lo, hi) of forward((x, adj)) -- inference only: the aggregation runs over that slice of the row pointers
        (gathering from all of x, or of x W^T), the dense part on hi - lo rows.  Row for row the arithmetic of forward(): a
        rank that answers a slice of the queries (ragraph_amd.sharded) needs only these rows before its retrieval can start."""
        g = as_csr(adj)
        rp = g.rowptr[lo:hi + 1]
        xs = sparse_features(x, probe=False)
        if xs is None and aggregate_first(x.shape[1], self.fc.weight.shape[0]):
            if x.is_cuda and K.spmm_linear_helps(hi - lo, x.shape[1], self.fc.weight.shape[0]):   # (one launch, same bits)
                return K.spmm_linear(rp, g.col, g.val, x, self.fc.weight, self.bias, act=K.ACT_PRELU, alpha=self._alpha())
            agg = K.spmm_csr(rp, g.col, g.val, x)
            return K.linear(agg, self.fc.weight, self.bias, act=K.ACT_PRELU, alpha=self._alpha())
        seq_fts = K.spmm_csr(xs[0], xs[1], xs[2], self._weight_t()) if xs is not None else K.linear(x, self.fc.weight)
        return K.spmm_csr(rp, g.col, g.val, seq_fts, bias=self.bias, act=K.ACT_PRELU, alpha=self._alpha())

    def forward(self, input, sparse=False):
        """input = (seq [n,F], adj): adj dense as in the reference (layers/gcn.py:26-40) or a CSRGraph.  The `sparse`
        flag is accepted for signature compatibility; aggregation is always the CSR SpMM kernel."""
        seq, adj = input[0], input[1]
        g = as_csr(adj)
        x = seq.squeeze(0) if seq.dim() == 3 else seq
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # fine-tuning (the few-shot flavours train this layer through decode()): the same kernels, with backward
            from .. import autograd as A
            seq_fts = A.linear(x, self.fc.weight)
            return A.spmm_csr(g, seq_fts, self.bias, K.ACT_PRELU, self.act.weight, self._alpha())
        xs = sparse_features(x, probe=False)
        if xs is None and aggregate_first(x.shape[1], self.fc.weight.shape[0]):
            # narrow features (c2: 128 -> 256): A_hat (X W^T) = (A_hat X) W^T, and the gathers of the aggregation -- what a
            # hop costs -- move half the bytes on the narrow side; bias + PReLU ride in the dense
            # kernel's epilogue.  Another association of the same sum (oracle/pipeline.py gcn_layer(order="aggregate_first")).
            if x.is_cuda and K.spmm_linear_helps(x.shape[0], x.shape[1], self.fc.weight.shape[0]):
                # one launch: the aggregated row is made in the dense kernel's LDS stage and never written (same bits)
                return K.spmm_linear(g.rowptr, g.col, g.val, x, self.fc.weight, self.bias, act=K.ACT_PRELU, alpha=self._alpha())
            if not g.has_long_rows and x.is_cuda and K.slices_help(x.shape[0], x.shape[1]):
                agg = K.spmm_csr_panels(g.rowptr, g.col, g.val, x, x_panels=False, y_panels=False)   # (same bits)
            else:
                agg = K.spmm_csr(g.rowptr, g.col, g.val, x, long_rows=g.has_long_rows)
            return K.linear(agg, self.fc.weight, self.bias, act=K.ACT_PRELU, alpha=self._alpha())
        if xs is not None:  # bag-of-words features: X W^T over X's non-zeros only -- the same bits (see above)
            seq_fts = K.spmm_csr(xs[0], xs[1], xs[2], self._weight_t())
        else:
            seq_fts = K.linear(x, self.fc.weight)                                                # :32
        return K.spmm_csr(g.rowptr, g.col, g.val, seq_fts, bias=self.bias, act=K.ACT_PRELU,      # :36-40 fused
                          alpha=self._alpha(), long_rows=g.has_long_rows)
