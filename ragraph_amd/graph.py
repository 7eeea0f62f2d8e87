"""CSR adjacency on the device: the hot path's input format.

The reference hands a DENSE n x n normalised adjacency to every layer (ragraph_utils/utility.py:45-69,
layers/gcn.py:36, Propagation.py:15-22): 40 GB per copy at n = 1e5.  The kernels take CSR (int64 rowptr, int32 col,
fp32 val); `as_csr` accepts what reference callers pass (a dense tensor) or a CSRGraph, and converts once.

Structure conversion: COO -> CSR (from_coo, transposed, permuted) runs on the library's own stable radix sort
(ragraph_coo_to_csr_i64) -- the edge flavour rebuilds its graph on every training step; only from_dense (the reference's
dense [n, n] input, converted once per graph) still enumerates the non-zeros with torch.nonzero.  Every floating-point
result comes from libragraph_hip.so.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import kernels as K


@dataclass
class CSRGraph:
    rowptr: torch.Tensor  # [n+1] int64
    col: torch.Tensor     # [nnz] int32
    val: torch.Tensor     # [nnz] fp32
    n: int
    _row_normalized: torch.Tensor | None = field(default=None, repr=False)
    _has_long_rows: bool | None = field(default=None, repr=False)
    _transposed: "CSRGraph | None" = field(default=None, repr=False)

    @property
    def shape(self):
        return (self.n, self.n)

    @property
    def nnz(self) -> int:
        return int(self.col.numel())

    @property
    def device(self):
        return self.val.device

    def squeeze(self, dim=0):  # reference code calls adj.squeeze(dim=0) on the dense tensor (layers/gcn.py:36)
        return self

    def cuda(self):  # reference code calls .cuda() on its inputs (layers/gcn.py:28-29)
        return self

    @property
    def has_long_rows(self) -> bool:
        """Does a row hold more than K.ROW_BLOCK edges (a hub of a power-law graph)?  Looked up once per graph (one
        reduction and read-back); the kernels then spread such rows over the chip instead of one lane group."""
        if self._has_long_rows is None:
            self._has_long_rows = bool(self.n > 0 and self.nnz > K.ROW_BLOCK and
                                       int((self.rowptr[1:] - self.rowptr[:-1]).max()) > K.ROW_BLOCK)
        return self._has_long_rows

    def row_normalized_values(self) -> torch.Tensor:
        """adj / adj.sum(1) on the non-zeros (Propagation.py:15-16), cached per graph."""
        if self._row_normalized is None:
            self._row_normalized = K.csr_row_normalize(self.rowptr, self.val)
        return self._row_normalized

    def transposed(self) -> "CSRGraph":
        """CSR of A^T (row j lists the i with A[i][j] != 0, ascending i), cached: what a backward pass through the SpMM
        multiplies by, and what PageRank pulls along."""
        if self._transposed is None:
            # (the edges are row-major already: a STABLE sort by column alone leaves every transposed row ascending)
            self._transposed, _ = CSRGraph.from_coo(self.col, self.row_ids(), self.val, self.n)
        return self._transposed

    def row_ids(self) -> torch.Tensor:
        """rows[e] = the row of CSR slot e (int64)."""
        if self.rowptr.is_cuda:
            return K.csr_row_ids(self.rowptr, self.nnz)
        return torch.repeat_interleave(torch.arange(self.n, device=self.device), self.rowptr[1:] - self.rowptr[:-1])

    # ---- locality ----------------------------------------------------------------------------------------------
    def permuted(self, order: torch.Tensor) -> "CSRGraph":
        """P A P^T: node order[i] of this graph becomes node i (rows AND columns), columns ascending within a row again.
        The SpMM over it gathers rows of X[order] -- neighbours that are close in `order` are close in memory, so a
        community-aware order (locality_order) turns Infinity-Cache gathers into L2 hits.  A row's terms are then summed
        in the NEW column order: the result equals the unpermuted one up to fp32 summation order (not bit for bit),
        which is why no forward applies this by itself -- the caller opts in, permutes X once and inverts at the end."""
        order = order.to(self.device, torch.int64)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(self.n, device=self.device)
        g, _ = CSRGraph.from_coo(inv[self.row_ids()], inv[self.col.long()], self.val, self.n, sort_cols=True)
        return g

    def locality_order(self) -> torch.Tensor:
        """Reverse Cuthill-McKee order of the (symmetrised) pattern -- structure-only bookkeeping, once per graph, on the
        host (scipy.sparse.csgraph): order[i] = the node that should be stored i-th."""
        import numpy as np
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        rp = self.rowptr.cpu().numpy()
        a = sp.csr_matrix((np.ones(self.nnz, dtype=np.int8), self.col.cpu().numpy(), rp), shape=(self.n, self.n))
        return torch.from_numpy(np.ascontiguousarray(reverse_cuthill_mckee(a, symmetric_mode=False)).astype(np.int64)).to(self.device)

    # ---- constructors ------------------------------------------------------------------------------------------
    @staticmethod
    def from_dense(adj: torch.Tensor) -> "CSRGraph":
        """Non-zeros of a dense [n,n] (or [1,n,n]) adjacency, rows ascending, columns ascending within a row."""
        a = adj.squeeze(0) if adj.dim() == 3 else adj
        n = a.shape[0]
        nz = torch.nonzero(a, as_tuple=False)  # row-major order
        rows, cols = nz[:, 0], nz[:, 1]
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=a.device)
        rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
        return CSRGraph(rowptr, cols.to(torch.int32), a[rows, cols].float().contiguous(), n)

    @staticmethod
    def from_coo(rows: torch.Tensor, cols: torch.Tensor, vals: torch.Tensor, n: int, sort_cols: bool = False):
        """COO -> CSR over `rows`, STABLE in the given edge order (the order scatter_add_ accumulates in,
        RAGraph_edge/modules/utils.py:17-32).  Returns (graph, perm): perm[e'] = original edge id of CSR slot e'."""
        rows = rows.to(torch.int64)
        if rows.is_cuda:
            # the product path: the library's own stable radix sort (csrc/ingest.hip: ragraph_coo_to_csr_i64) -- this runs on
            # EVERY training step of the edge flavour (a re-drawn edge set), for every training SpMM's transposed pattern and in
            # the gather backward; no read-back, nothing from another library
            rowptr, col, perm = K.coo_to_csr(rows, cols, n, sort_cols=sort_cols)
            return CSRGraph(rowptr, col, vals[perm].float().contiguous(), n), perm
        # host tensors (the CPU host-logic tests): the same construction with torch ops
        key = rows * n + cols.to(torch.int64) if sort_cols else rows
        perm = torch.sort(key, stable=True).indices
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=rows.device)
        rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
        g = CSRGraph(rowptr, cols[perm].to(torch.int32).contiguous(), vals[perm].float().contiguous(), n)
        return g, perm

    @staticmethod
    def from_edge_index_sym_normalized(edge_index: torch.Tensor, n: int) -> "CSRGraph":
        """D^-1/2 (A + I) D^-1/2 directly in CSR -- what ragraph_utils/utility.py:19-26,45-66 builds densely through
        scipy (coo_matrix of ones -> todense sums duplicate edges; + eye; normalize_adj).  The normalisation is done
        in float64 and cast to fp32 last, as scipy + torch.FloatTensor do."""
        dev = edge_index.device
        if dev.type == "cuda":  # the product path: one sort + a few kernels on the device (csrc/ingest.hip)
            rowptr, col, val = K.csr_sym_normalized_from_edges(edge_index, n)
            return CSRGraph(rowptr, col, val, n)
        # host tensors (the CPU host-logic tests): the same construction with torch ops
        loops = torch.arange(n, device=dev, dtype=torch.int64)
        r = torch.cat([edge_index[0].to(torch.int64), loops])
        c = torch.cat([edge_index[1].to(torch.int64), loops])
        key, counts = torch.unique(r * n + c, return_counts=True)  # sorted: row-major, ascending columns
        r, c = key // n, key % n
        vals = counts.double()
        deg = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, r, vals)
        dinv = deg.pow(-0.5)
        dinv[torch.isinf(dinv)] = 0.0
        # normalize_adj (utility.py:19-26) returns (A D)^T D = D A^T D: entry (i,j) = d_i * A[j,i] * d_j, with d from A's
        # ROW sums.  Emit the transposed pattern so a directed edge list gives the reference's matrix too.
        v = (vals * dinv[r]) * dinv[c]
        order = torch.sort(c * n + r).indices
        out_r, out_c, v = c[order], r[order], v[order]
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(out_r, minlength=n), 0)
        return CSRGraph(rowptr, out_c.to(torch.int32), v.float().contiguous(), n)


def as_csr(adj) -> CSRGraph:
    """Accept what the reference call surface passes (dense [n,n] / [1,n,n] tensor) or a CSRGraph."""
    if isinstance(adj, CSRGraph):
        return adj
    if isinstance(adj, torch.Tensor):
        if adj.layout == torch.sparse_csr:
            return CSRGraph(adj.crow_indices().to(torch.int64), adj.col_indices().to(torch.int32),
                            adj.values().float().contiguous(), adj.shape[-1])
        return CSRGraph.from_dense(adj)
    raise TypeError(f"adjacency must be a dense tensor, a sparse CSR tensor or a CSRGraph, not {type(adj)}")
