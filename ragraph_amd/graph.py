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


class TilePlan:
    """CSRGraph.tile_plan: wp [16 C + 1] int32 (first wave-batch of every (chunk, wave)), col3 int32 / row3 int16 [slots] (64 per
    wave-batch; row -1 = 0xFFFF = no edge), perm [nnz] (plan order -> CSR slot), slot [nnz] (plan order -> slot)."""

    def __init__(self, wp, col3, row3, perm, slot, slots, RG, S, SB, C, passes):
        self.wp, self.col3, self.row3, self.perm, self.slot, self.slots = wp, col3, row3, perm, slot, int(slots)
        self.RG, self.S, self.SB, self.C, self.passes = int(RG), int(S), int(SB), int(C), int(passes)
        self._val_tag, self._val2 = None, None


@dataclass
class CSRGraph:
    rowptr: torch.Tensor  # [n+1] int64
    col: torch.Tensor     # [nnz] int32
    val: torch.Tensor     # [nnz] fp32
    n: int
    _row_normalized: torch.Tensor | None = field(default=None, repr=False)
    _has_long_rows: bool | None = field(default=None, repr=False)
    _transposed: "CSRGraph | None" = field(default=None, repr=False)
    _tile_plans: dict = field(default_factory=dict, repr=False)

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

    # ---- graph tiling (csrc/sparse.hip: spmm_tiled_kernel) -------------------------------------------------------------
    TILE_SOURCE_BYTES = int(__import__("os").environ.get("RAGRAPH_SPMM_TILE_SOURCE_BYTES", str(5 << 18)))   # 1.25 MiB (c2: 114 us per hop; 2.5 MiB: 118)

    TILE_NEAR_ROWS = 4096

    def tile_plan(self, panels: int):
        """The plan of the graph-tiled hop for features of `panels` 32-column panels, made once per graph (cached; None when
        the graph cannot or should not be tiled: a row longer than ROW_BLOCK edges, columns that do not ascend inside a row -- the
        tiled kernel consumes a row's edges source block by source block, which is the CSR order only then --, a numbering
        that already keeps most neighbours within TILE_NEAR_ROWS rows, skewed degrees, or host tensors).
        Destination rows: C chunks of RC = 128 RG rows (RG <= 9: the chunk's 32-column sums fill a workgroup's LDS), as few
        passes over the chunks as that allows; source rows: S blocks of at most TILE_SOURCE_BYTES of 128-byte lines; edges:
        one stable sort (the library's own) by (chunk, 8-lane group, source block) -- inside a bucket the CSR order stays."""
        share = 1 if panels >= 8 else 8 // max(panels, 1)
        if share in self._tile_plans:
            return self._tile_plans[share]
        plan = None
        if self.rowptr.is_cuda and self.n >= 1 and self.nnz >= 1 and self.nnz < 2 ** 31 - 16 and not self.has_long_rows:
            rows = self.row_ids()
            col = self.col.long()
            unsorted = bool(((col[1:] < col[:-1]) & (rows[1:] == rows[:-1])).any()) if self.nnz > 1 else False
            # a numbering that already keeps neighbours close (a community graph after locality_order(), a banded matrix) is
            # served from L1 / L2 by the panel kernel, whose per-edge cost is lower: measured on the reordered community graph
            # of bench.py's gnn_fwd.structured_graph, forward 0.492 ms on the panel kernels, 0.556 tiled; c2's graph (self loop
            # + ring: 27 % of the edges near) 0.545 / 0.534.  Tiled only when fewer than half of the edges are near.
            near = float(((col - rows).abs() <= self.TILE_NEAR_ROWS).float().mean()) if self.nnz else 1.0
            if not unsorted and near < 0.5:
                NG, WPX = 128, 32
                passes = 1
                while -(-self.n // (WPX * share * passes * NG)) > 9:
                    passes += 1
                RG = -(-self.n // (WPX * share * passes * NG))
                RC = RG * NG
                C = -(-self.n // RC)
                S = max(1, -(-self.n * 128 // self.TILE_SOURCE_BYTES))
                SB = -(-self.n // S)
                bucket = ((rows // RC) * NG + (rows % RC) // RG) * S + col // SB
                gp, col2, perm = K.coo_to_csr(bucket, col, C * NG * S)
                # wave-batches: the eight groups of a wave keep their batch's words in ONE 64-slot record (slot 8 g + i = edge
                # i of group g's batch); a wave walks as many batches as its longest group's run needs, the others' tails are
                # NULL slots (row 0xFFFF, column 0)
                grp_ptr = gp[::S].contiguous()                                   # [C * NG + 1]: the groups' runs
                run0 = grp_ptr[:-1]
                run_len = grp_ptr[1:] - run0
                nb_wave = ((run_len + 7) // 8).view(C * 16, 8).max(dim=1).values   # batches per (chunk, wave)
                wp = torch.zeros(C * 16 + 1, dtype=torch.int64, device=self.device)
                wp[1:] = torch.cumsum(nb_wave, 0)
                slots = (int(wp[-1]) + 1) * 64                                   # (+ one NULL wave-batch behind the last)
                if slots > 1.5 * self.nnz + 64 * 16 * C:   # skewed degrees: a wave walks its longest group's run -- too many NULL slots
                    self._tile_plans[share] = None
                    return None
                grp = K.csr_row_ids(grp_ptr, self.nnz)                           # group of every edge (plan order)
                k = torch.arange(self.nnz, device=self.device) - run0[grp]
                slot = (wp[grp // 8] + k // 8) * 64 + (grp % 8) * 8 + k % 8
                col3 = torch.zeros(slots, dtype=torch.int32, device=self.device)
                row3 = torch.full((slots,), -1, dtype=torch.int16, device=self.device)   # 0xFFFF
                col3[slot] = col2
                row3[slot] = (rows[perm] % RC).to(torch.int16)
                plan = TilePlan(wp.to(torch.int32), col3, row3, perm, slot, slots, RG, S, SB, C, passes)
        self._tile_plans[share] = plan
        return plan

    def tiled_values(self, plan, val: torch.Tensor) -> torch.Tensor:
        """The edge values in the plan's slots (0 in the NULL slots), cached per value tensor and version."""
        tag = (val.data_ptr(), val._version, val.numel())
        if plan._val_tag != tag:
            v3 = torch.zeros(plan.slots, dtype=torch.float32, device=val.device)
            v3[plan.slot] = val[plan.perm]
            plan._val2, plan._val_tag = v3, tag
        return plan._val2

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
