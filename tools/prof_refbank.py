#!/usr/bin/env python3
"""The 100 000-query retrieval against the bank the reference's own recipe builds (bench.py's retrieval_reference_bank.B100000),
repeated: for rocprofv3 kernel stats.   python tools/prof_refbank.py [reps]"""
import os, sys, types, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from ragraph_amd.data import synthetic_big_graph
from ragraph_amd.graph import CSRGraph

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
args = types.SimpleNamespace(feat=128, dim=256, classes=3, k=10, nodes=100_000, bank=1_000_000)
adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(args.nodes, 10, seed=8, device=dev), args.nodes)
feats = torch.randn(args.nodes, args.feat, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
print(bench.reference_bank_rates(args, dev, adj, feats, batches=(100_000,), reps=reps))
