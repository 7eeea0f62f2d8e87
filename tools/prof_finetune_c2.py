#!/usr/bin/env python3
"""A few fine-tuning steps of the node flavour at c2, for rocprofv3 (kernel stats of finetune_step.node_c2)."""
import os, sys, types, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
import bench, bench_blocks as BB

args = types.SimpleNamespace(feat=128, dim=256, classes=3, k=10, nodes=100_000, bank=1_000_000, emulate_rank_of=0, key_shards=2)
dev = torch.device("cuda", 0)
model, feats, adj, _ = bench.build_workload(args, dev, 0, 1, "keys")
with torch.no_grad():
    for _ in range(6):
        model(feats, adj)
torch.cuda.synchronize()
labels = torch.randint(0, 3, (feats.shape[0],), device=dev)
params = [p for p in model.parameters() if p.requires_grad]
model.train()
opt = torch.optim.Adam(params, lr=1e-3)
ms = BB.event_ms(lambda: BB._node_step_gpu(model, feats, adj, labels, opt), int(sys.argv[1]) if len(sys.argv) > 1 else 8, warm=2)
print(f"fine-tuning step {ms:.3f} ms")
