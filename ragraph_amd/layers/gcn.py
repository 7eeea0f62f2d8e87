"""Mirror of RAGraph_*/layers/gcn.py: one GCN layer PReLU(A_hat (X W^T) + b), same parameter names (fc, act, bias)."""
import torch
import torch.nn as nn

from .. import kernels as K
from ..graph import as_csr


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

    def _alpha(self) -> float:
        """PReLU slope as a host scalar (a kernel argument), re-read only when the parameter changes."""
        w = self.act.weight
        tag = (w.data_ptr(), w._version)
        if self._alpha_cache[0] != tag:
            self._alpha_cache = (tag, float(w.detach().reshape(-1)[0]))
        return self._alpha_cache[1]

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
        seq_fts = K.linear(x, self.fc.weight)                                                    # :32
        return K.spmm_csr(g.rowptr, g.col, g.val, seq_fts, bias=self.bias, act=K.ACT_PRELU,      # :36-40 fused
                          alpha=self._alpha(), long_rows=g.has_long_rows)
