#!/usr/bin/env python3
"""SpMM locality experiment (c2 shape: 100k nodes, mean degree 10, D = 256 -> 0.214 GB algorithmic bytes per hop):
one hop of A @ X on  (a) the c2 Erdos-Renyi graph, (b) a community-structured graph in its shuffled numbering,
(c) the same after CSRGraph.locality_order() (reverse Cuthill-McKee), (d) in the generator's ground-truth community order.
   python tools/spmm_locality.py MODE [reps]     MODE in random | shuffled | rcm | truth
Run under rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE) by tools/gpu_spmm_locality.sh."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragraph_amd import kernels as K
from ragraph_amd.data import synthetic_big_graph, synthetic_community_graph
from ragraph_amd.graph import CSRGraph

mode = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
n, D = 100_000, 256
torch.manual_seed(0)
X = torch.randn(n, D, device=dev)
if mode == "random":
    g = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, device=dev), n)
    order = None
else:
    ei, member = synthetic_community_graph(n, 10, 512, 0.9, device=dev)
    g = CSRGraph.from_edge_index_sym_normalized(ei, n)
    t0 = time.perf_counter()
    order = {"shuffled": None, "rcm": g.locality_order() if mode == "rcm" else None,
             "truth": torch.sort(member.to(dev), stable=True).indices}[mode]
    t_order = time.perf_counter() - t0
if order is not None:
    base = K.spmm_csr(g.rowptr, g.col, g.val, X)
    g = g.permuted(order)
    X = X[order].contiguous()
    got = K.spmm_csr(g.rowptr, g.col, g.val, X)
    err = (got - base[order]).abs().max().item()
    print(f"reordered result vs natural order: max abs diff {err:.2e} (summation order only); ordering took {t_order:.2f} s")
    assert err < 1e-4
# mean |i - j| over the edges, in rows of X: what the gather's footprint looks like
rows = torch.repeat_interleave(torch.arange(n, device=dev), g.rowptr[1:] - g.rowptr[:-1])
span = (rows - g.col.long()).abs().float().mean().item()
for _ in range(3):
    K.spmm_csr(g.rowptr, g.col, g.val, X)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    K.spmm_csr(g.rowptr, g.col, g.val, X)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1000
alg = (n * D * 4 * 2 + g.nnz * 8) / 1e9
print(f"mode={mode} n={n} nnz={g.nnz} D={D}: {us:.1f} us per hop, mean |row - col| = {span:.0f}, algorithmic {alg:.3f} GB "
      f"-> {alg / us * 1e3:.2f} TB/s effective")
