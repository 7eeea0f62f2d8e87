"""Bank-side dispatch of the exact top-k (torch only: importable on a CPU box for the gloo tests)."""
from __future__ import annotations

import torch


class KeyIndex:
    """One bank version on the device: the normalised keys plus the copies the faster kernels stream (packed fp32 for
    the LDS-DMA ring, bf16 for the filter), made on first use, and the dispatch to the fastest exact top-k for a batch.
    `ops` supplies the kernels (default: this module); an object without the optional entries (the CPU tests' oracle
    shim) simply always takes topk_cosine."""

    MAX_FILTERED_BATCH = 262144

    def __init__(self, keys_normalized: torch.Tensor, ops=None):
        if ops is None:
            from . import kernels as ops  # the HIP library; raises loudly without a GPU
        self.ops = ops
        self.keys_normalized = keys_normalized
        self._packed = None
        self._bf16 = None
        self._filter_off = False  # set when this bank defeats the filter (see topk)

    def topk(self, q: torch.Tensor, k: int, idx_base: int = 0):
        ops, kn = self.ops, self.keys_normalized
        B, D = q.shape
        fhelps = getattr(ops, "filter_helps", None)
        if fhelps is not None and not self._filter_off and fhelps(B, kn.shape[0], D, k):
            if self._bf16 is None:
                self._bf16 = ops.keys_to_bf16(kn)
            # (the packed fp32 copy is not made for this path: its first bound comes from the bf16 copy itself)
            if B > self.MAX_FILTERED_BATCH:
                # the candidate lists take 8 KiB per query: slabs of 256 k queries keep the workspace at 2 GiB (the edge
                # flavour asks for all 4 M nodes at once) at the throughput of one big call
                outs = [self.topk(q[b0:b0 + self.MAX_FILTERED_BATCH], k, idx_base)
                        for b0 in range(0, B, self.MAX_FILTERED_BATCH)]
                return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
            s, i, n_over = ops.topk_cosine_filtered(q, kn, self._bf16, k, idx_base=idx_base, keys_packed=self._packed)
            # A bank of near-duplicates (thousands of keys within the bf16 bound of a query's k-th best) overflows the
            # candidate lists, and every such row is recomputed with the fp32 kernels: still exact, but once a quarter
            # of a sizeable batch goes that way the filter only adds its own cost -- this bank version stays on fp32.
            if B >= 64 and 4 * n_over > B:
                self._filter_off = True
            return s, i
        helps = getattr(ops, "packed_keys_help", None)
        if helps is not None and helps(B, D, k):
            if self._packed is None:
                self._packed = ops.pack_keys(kn)
            return ops.topk_cosine(q, kn, k, idx_base=idx_base, keys_packed=self._packed)
        return ops.topk_cosine(q, kn, k, idx_base=idx_base)
