#!/usr/bin/env python3
"""The node flavour's fine-tuning step at c2 (bench.py's finetune_step.node_c2) next to the inference forward on the same
model, several repetitions:   RAGRAPH_SPMM_LINEAR=0|1 python tools/finetune_c2_ab.py"""
import os, sys, types, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
import bench, bench_blocks as BB

args = types.SimpleNamespace(feat=128, dim=256, classes=3, k=10, nodes=100_000, bank=1_000_000, emulate_rank_of=0, key_shards=2)
dev = torch.device("cuda", 0)
model, feats, adj, _ = bench.build_workload(args, dev, 0, 1, "keys")
with torch.no_grad():
    fwd = BB.event_ms(lambda: model(feats, adj), 10, warm=5)
print(f"RAGRAPH_SPMM_LINEAR={os.environ.get('RAGRAPH_SPMM_LINEAR', '1')}: inference forward {fwd:.3f} ms")
for i in range(3):
    rec = BB.finetune_node(dev, 1, "c2", c2=(model, feats, adj), cpu=False)
    print(f"  fine-tuning step {rec['ms']:.3f} ms")
with torch.no_grad():
    fwd = BB.event_ms(lambda: model(feats, adj), 10, warm=2)
print(f"  inference forward again {fwd:.3f} ms")
