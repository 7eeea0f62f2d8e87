"""One propagation hop at c2's shape (n = 100 000, D = 256, panel-major in and out) on the panel kernel and on the graph-tiled
kernel, repeated: run under rocprofv3 --pmc (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, separate passes) for the bytes
behind the L2s and the L2 hit rate of each.     python tools/prof_spmm_hop.py [reps] [source_block_bytes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragraph_amd import kernels as K
from ragraph_amd.data import synthetic_big_graph
from ragraph_amd.graph import CSRGraph

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if len(sys.argv) > 2:
    CSRGraph.TILE_SOURCE_BYTES = int(sys.argv[2])
dev = torch.device("cuda:0")
n, D = 100_000, 256
g = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, seed=8, device=dev), n)
vn = K.csr_row_normalize(g.rowptr, g.val)
Hp = torch.randn(n, D, device=dev)
plan = g.tile_plan(D // 32)
v2 = g.tiled_values(plan, vn)


def t(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


a = t(lambda: K.spmm_csr_panels(g.rowptr, g.col, vn, Hp, True, True, act=K.ACT_RELU))
b = t(lambda: K.spmm_csr_tiled(plan, v2, Hp, n, True, True, act=K.ACT_RELU))
print(f"hop n={n} D={D} nnz={g.nnz}: panel kernel {a:.1f} us, tiled kernel {b:.1f} us (RG {plan.RG}, passes {plan.passes}, S {plan.S}, "
      f"{plan.slots} slots for {g.nnz} edges)")
