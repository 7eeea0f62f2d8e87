"""Mirror of RAGraph_*/ragraph_utils/Propagation.py."""
import torch

from .. import kernels as K
from ..graph import as_csr


class Propagation:
    @staticmethod
    def aggregate_k_hop_features(adj, x: torch.Tensor, k: int, rows=None) -> torch.Tensor:
        """k x { x = relu((adj / adj.sum(1)) @ x) } -- Propagation.py:7-27.  `adj` dense (reference form) or CSRGraph.
        Each hop is one CSR SpMM with the ReLU fused into the store; the row normalisation is cached on the graph.
        rows = (lo, hi): only rows [lo, hi) of the result are wanted (a rank's slice of the nodes, ragraph_amd.sharded):
        the LAST hop runs over those rows alone -- every earlier hop feeds all rows of the next -- same bits per row."""
        g = as_csr(adj)
        if k <= 0:
            return x if rows is None else x[rows[0]:rows[1]].contiguous()
        valn = g.row_normalized_values()
        if (not (torch.is_grad_enabled() and x.requires_grad) and x.is_cuda and x.shape[1] % 32 == 0
                and K.tiles_help(x.shape[0], x.shape[1])):
            # large graphs, round 6: the GRAPH is tiled (once; CSRGraph.tile_plan) so that what an XCD gathers from at any
            # moment fits its L2 -- and the features stay panel-major between the hops as below; same bits.  The last hop of a
            # rank's row slice takes the panel kernel over that slice of the row pointers.
            plan = g.tile_plan(x.shape[1] // 32)
            if plan is not None:
                val2 = g.tiled_values(plan, valn)
                for hop in range(int(k)):
                    last = hop == int(k) - 1
                    if last and rows is not None:
                        return K.spmm_csr_panels(g.rowptr[rows[0]:rows[1] + 1], g.col, valn, x, x_panels=hop > 0, y_panels=False,
                                                 act=K.ACT_RELU)
                    x = K.spmm_csr_tiled(plan, val2, x, g.n, x_panels=hop > 0, y_panels=not last, act=K.ACT_RELU)
                return x
        if (not g.has_long_rows and not (torch.is_grad_enabled() and x.requires_grad) and x.is_cuda
                and K.panels_help(x.shape[0], x.shape[1], int(k))):
            # large graphs: the features stay panel-major between the hops (an XCD gathers one 128-byte line per neighbour
            # from its own panel instead of 1-KiB rows from the whole table); same bits.  rows: the last hop over that
            # slice of the row pointers only.
            for hop in range(int(k)):
                last = hop == int(k) - 1
                rp = g.rowptr[rows[0]:rows[1] + 1] if (last and rows is not None) else g.rowptr
                x = K.spmm_csr_panels(rp, g.col, valn, x, x_panels=hop > 0, y_panels=not last, act=K.ACT_RELU)
            return x
        for hop in range(int(k)):
            if rows is not None and hop == int(k) - 1 and not g.has_long_rows:
                x = K.spmm_csr(g.rowptr[rows[0]:rows[1] + 1], g.col, valn, x, act=K.ACT_RELU)
                return x
            x = K.spmm_csr(g.rowptr, g.col, valn, x, act=K.ACT_RELU, long_rows=g.has_long_rows)
        return x if rows is None else x[rows[0]:rows[1]].contiguous()
