"""Mirror of RAGraph_*/ragraph_utils/Propagation.py."""
import torch

from .. import kernels as K
from ..graph import as_csr


class Propagation:
    @staticmethod
    def aggregate_k_hop_features(adj, x: torch.Tensor, k: int) -> torch.Tensor:
        """k x { x = relu((adj / adj.sum(1)) @ x) } -- Propagation.py:7-27.  `adj` dense (reference form) or CSRGraph.
        Each hop is one CSR SpMM with the ReLU fused into the store; the row normalisation is cached on the graph."""
        g = as_csr(adj)
        if k <= 0:
            return x
        valn = g.row_normalized_values()
        for _ in range(int(k)):
            x = K.spmm_csr(g.rowptr, g.col, valn, x, act=K.ACT_RELU, long_rows=g.has_long_rows)
        return x
