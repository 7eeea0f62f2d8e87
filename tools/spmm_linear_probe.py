#!/usr/bin/env python3
"""The aggregate-first encoder of c2 (100k nodes, 128 -> 256) as one launch (ragraph_spmm_linear_f32) against the two launches
it replaces; a rank-of-8's 12 500-row slice; the whole GNN forward both ways."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K
from ragraph_amd.data import synthetic_big_graph
from ragraph_amd.graph import CSRGraph

dev = torch.device("cuda", 0)
n, F, D = 100_000, 128, 256
g = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, seed=8, device=dev), n)
X = torch.randn(n, F, device=dev)
W = torch.randn(D, F, device=dev) * 0.05
b = torch.randn(D, device=dev) * 0.1


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


two = lambda: K.linear(K.spmm_csr_panels(g.rowptr, g.col, g.val, X, x_panels=False, y_panels=False), W, b, act=K.ACT_PRELU, alpha=0.25)
one = lambda: K.spmm_linear(g.rowptr, g.col, g.val, X, W, b, act=K.ACT_PRELU, alpha=0.25)
print("same bits:", bool(torch.equal(two(), one())))
print(f"two launches (panel spmm + linear): {t(two):7.1f} us")
print(f"one launch:                         {t(one):7.1f} us")
for rows in (12_500, 25_000, 50_000):
    rp = g.rowptr[1000:1000 + rows + 1]
    print(f"{rows} rows: two {t(lambda: K.linear(K.spmm_csr(rp, g.col, g.val, X), W, b, act=K.ACT_PRELU, alpha=0.25)):7.1f} us, "
          f"one {t(lambda: K.spmm_linear(rp, g.col, g.val, X, W, b, act=K.ACT_PRELU, alpha=0.25)):7.1f} us")
X64 = torch.randn(n, 64, device=dev)
W64 = torch.randn(D, 64, device=dev) * 0.05
print(f"K=64: two {t(lambda: K.linear(K.spmm_csr(g.rowptr, g.col, g.val, X64), W64, b, act=K.ACT_PRELU, alpha=0.25)):7.1f} us, "
      f"one {t(lambda: K.spmm_linear(g.rowptr, g.col, g.val, X64, W64, b, act=K.ACT_PRELU, alpha=0.25)):7.1f} us")
