#!/usr/bin/env python3
"""Kernel times of the c2 GNN forward's pieces (100k-node graph, F = 128 -> 256): SpMM at widths 128 / 256, the dense layer
with and without the fused bias + PReLU, the two associations of the encoder, a propagation hop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K
from ragraph_amd.data import synthetic_big_graph
from ragraph_amd.graph import CSRGraph

dev = torch.device("cuda", 0)
n, F, D = 100_000, 128, 256
g = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, seed=8, device=dev), n)
X = torch.randn(n, F, device=dev)
H = torch.randn(n, D, device=dev)
W = torch.randn(D, F, device=dev) * 0.05
b = torch.randn(D, device=dev) * 0.1
vn = K.csr_row_normalize(g.rowptr, g.val)


def t(fn, reps=30):
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


print(f"spmm D=128 (no epilogue):          {t(lambda: K.spmm_csr(g.rowptr, g.col, g.val, X)):7.1f} us")
print(f"spmm D=256 (bias + PReLU):         {t(lambda: K.spmm_csr(g.rowptr, g.col, g.val, H, bias=b, act=K.ACT_PRELU, alpha=0.25)):7.1f} us")
print(f"spmm D=256 (ReLU, a hop):          {t(lambda: K.spmm_csr(g.rowptr, g.col, vn, H, act=K.ACT_RELU)):7.1f} us")
print(f"linear 128 -> 256:                 {t(lambda: K.linear(X, W)):7.1f} us")
print(f"linear 128 -> 256 + bias + PReLU:  {t(lambda: K.linear(X, W, b, act=K.ACT_PRELU, alpha=0.25)):7.1f} us")
print(f"encoder, reference order:          {t(lambda: K.spmm_csr(g.rowptr, g.col, g.val, K.linear(X, W), bias=b, act=K.ACT_PRELU, alpha=0.25)):7.1f} us")
print(f"encoder, aggregate first:          {t(lambda: K.linear(K.spmm_csr(g.rowptr, g.col, g.val, X), W, b, act=K.ACT_PRELU, alpha=0.25)):7.1f} us")
# the XCD-sliced kernel in every layout combination (a hop at D = 256; the narrow aggregation at D = 128)
Hp = H.view(n, D // 32, 32).permute(1, 0, 2).contiguous().view(n, D)
for xp, yp in ((0, 0), (0, 1), (1, 1), (1, 0)):
    src = Hp if xp else H
    print(f"sliced hop D=256, x_panels={xp} y_panels={yp}: {t(lambda: K.spmm_csr_panels(g.rowptr, g.col, vn, src, bool(xp), bool(yp), act=K.ACT_RELU)):7.1f} us")
print(f"sliced spmm D=128 row -> row:      {t(lambda: K.spmm_csr_panels(g.rowptr, g.col, g.val, X, False, False)):7.1f} us")
from ragraph_amd.ragraph_utils import Propagation
os.environ["RAGRAPH_SPMM_TILED"] = "0"
print(f"3 hops (panel kernels):            {t(lambda: Propagation.aggregate_k_hop_features(g, H, 3)):7.1f} us")
os.environ["RAGRAPH_SPMM_TILED"] = "1"

# round 6: the graph-tiled kernel (CSRGraph.tile_plan) in the same layout combinations, and the source-block size
for sb in (5 << 19, 5 << 18, 13 << 20):
    CSRGraph.TILE_SOURCE_BYTES = sb
    g._tile_plans.clear()
    plan = g.tile_plan(D // 32)
    v2 = g.tiled_values(plan, vn)
    for xp, yp in ((1, 1), (0, 1), (1, 0), (0, 0)):
        src = Hp if xp else H
        print(f"tiled hop D=256 (RG {plan.RG}, passes {plan.passes}, S {plan.S}: {sb >> 10} KiB blocks), x_panels={xp} y_panels={yp}: "
              f"{t(lambda: K.spmm_csr_tiled(plan, v2, src, n, bool(xp), bool(yp), act=K.ACT_RELU)):7.1f} us")
    plan4 = g.tile_plan(F // 32)
    v4 = g.tiled_values(plan4, g.val)
    print(f"tiled spmm D=128 row -> row (RG {plan4.RG}, passes {plan4.passes}, S {plan4.S}): {t(lambda: K.spmm_csr_tiled(plan4, v4, X, n, False, False)):7.1f} us")
CSRGraph.TILE_SOURCE_BYTES = 5 << 18
g._tile_plans.clear()
print(f"3 hops (product path):             {t(lambda: Propagation.aggregate_k_hop_features(g, H, 3)):7.1f} us")
