import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd.data import synthetic_big_graph
from ragraph_amd.graph import CSRGraph
from ragraph_amd.preprompt import PrePrompt
from ragraph_amd.RAGraph import RAGraph
dev = torch.device("cuda:0"); torch.manual_seed(0)
n, F, D, C = 2708, 1433, 128, 7
adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 4, seed=7, device=dev), n)
X = (torch.rand(n, F, device=dev) < 0.0127).float(); X = X / X.sum(1, keepdim=True).clamp_min(1)
m = RAGraph(PrePrompt(F, D, "prelu", 1, 0.3).to(dev), None, F, C, D, device=dev).eval()
m.toy_graph_base.retrieve_num = 5
m.toy_graph_base.add_resources(torch.nn.functional.normalize(torch.randn(10000, D, device=dev), dim=-1), torch.randn(10000, D, device=dev), torch.nn.functional.one_hot(torch.randint(0, C, (10000,), device=dev), C).float())
with torch.no_grad():
    for _ in range(20): m(X, adj)
torch.cuda.synchronize()
