"""Bank-side dispatch of the exact top-k (torch only: importable on a CPU box for the gloo tests)."""
from __future__ import annotations

import torch


class KeyIndex:
    """One bank version on the device: the normalised keys plus the copies the faster kernels stream (packed fp32 for
    the LDS-DMA ring, bf16 for the filter), made on first use, and the dispatch to the fastest exact top-k for a batch.
    `ops` supplies the kernels (default: this module); an object without the optional entries (the CPU tests' oracle
    shim) simply always takes topk_cosine."""

    def __init__(self, keys_normalized: torch.Tensor, ops=None):
        if ops is None:
            from . import kernels as ops  # the HIP library; raises loudly without a GPU
        self.ops = ops
        self.keys_normalized = keys_normalized
        self._packed = None
        self._bf16 = None

    def topk(self, q: torch.Tensor, k: int, idx_base: int = 0):
        ops, kn = self.ops, self.keys_normalized
        B, D = q.shape
        kp = None
        helps = getattr(ops, "packed_keys_help", None)
        if helps is not None and helps(B, D, k):
            if self._packed is None:
                self._packed = ops.pack_keys(kn)
            kp = self._packed
        fhelps = getattr(ops, "filter_helps", None)
        if fhelps is not None and fhelps(B, kn.shape[0], D, k):
            if self._bf16 is None:
                self._bf16 = ops.keys_to_bf16(kn)
            s, i, _ = ops.topk_cosine_filtered(q, kn, self._bf16, k, idx_base=idx_base, keys_packed=kp)
            return s, i
        if kp is not None:
            return ops.topk_cosine(q, kn, k, idx_base=idx_base, keys_packed=kp)
        return ops.topk_cosine(q, kn, k, idx_base=idx_base)
