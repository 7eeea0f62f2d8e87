"""The GNN forward of the c2 step alone (1 GCN layer + 3 propagation hops on the 100 000-node graph), repeated: run under
rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / WRITE_SIZE) for bench.py's gnn_fwd.bytes_counter.
    python tools/prof_gnn.py [reps] [structured]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragraph_amd.data import synthetic_big_graph, synthetic_community_graph
from ragraph_amd.graph import CSRGraph
from ragraph_amd.preprompt import PrePrompt
from ragraph_amd.ragraph_utils import Propagation

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
structured = len(sys.argv) > 2 and sys.argv[2] == "structured"
dev = torch.device("cuda:0")
n, F, D, hops = 100_000, 128, 256, 3
torch.manual_seed(0)
pre = PrePrompt(F, D, "prelu", 1, 0.3).to(dev)
if structured:
    ei, _ = synthetic_community_graph(n, 10, 512, 0.9, device=dev)
    adj = CSRGraph.from_edge_index_sym_normalized(ei, n)
    adj = adj.permuted(adj.locality_order())
else:
    adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, seed=8, device=dev), n)
X = torch.randn(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
with torch.no_grad():
    for _ in range(3):
        Propagation.aggregate_k_hop_features(adj, pre.inference(X, adj), hops)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        Propagation.aggregate_k_hop_features(adj, pre.inference(X, adj), hops)
    e1.record()
    torch.cuda.synchronize()
print(f"gnn_forward n={n} F={F} D={D} hops={hops} nnz={adj.nnz}{' structured+reordered' if structured else ''}: "
      f"{e0.elapsed_time(e1) / reps:.4f} ms per forward")
