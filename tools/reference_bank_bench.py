#!/usr/bin/env python3
"""Retrieval against a bank made by the reference's own recipe (ToyGraphBase.py:91-119 + Augmentation.py:9-20: ~75 % of
the rows one vector, sampled rows repeated), before / after the exact-duplicate collapsing of KeyIndex: which path the
dispatch ends on and queries/s at B = 1 / 500 / 100 000 (bench.py::reference_bank_rates, dedup on and off).
    python tools/reference_bank_bench.py [--bank 1000000] [--skip-uncollapsed]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    skip = "--skip-uncollapsed" in sys.argv
    sys.argv = [a for a in sys.argv if a != "--skip-uncollapsed"]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.graph import CSRGraph

    adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(args.nodes, 10, seed=8, device=dev), args.nodes)
    feats = torch.randn(args.nodes, args.feat, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
    print("collapsed  :", json.dumps(bench.reference_bank_rates(args, dev, adj, feats)), flush=True)
    if not skip:
        # (small batches first: a fresh index that meets 100 000 overflowing queries at once repairs every one of them by
        # an exact scan before the count can take the bank off the filter)
        print("uncollapsed:", json.dumps(bench.reference_bank_rates(args, dev, adj, feats, dedup=False)), flush=True)


if __name__ == "__main__":
    main()
