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
        for hop in range(int(k)):
            if rows is not None and hop == int(k) - 1 and not g.has_long_rows:
                x = K.spmm_csr(g.rowptr[rows[0]:rows[1] + 1], g.col, valn, x, act=K.ACT_RELU)
                return x
            x = K.spmm_csr(g.rowptr, g.col, valn, x, act=K.ACT_RELU, long_rows=g.has_long_rows)
        return x if rows is None else x[rows[0]:rows[1]].contiguous()
