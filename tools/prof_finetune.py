"""The node flavour's fine-tuning step on the c2 shape (finetune-rag.py:77-84: forward in train mode + cross entropy + backward
+ Adam), repeated: run under rocprofv3 --kernel-trace --stats for the per-kernel split of a step.
    python tools/prof_finetune.py [steps]"""
import os
import sys
import types

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
import torch

import bench
import bench_blocks as BB

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
args = types.SimpleNamespace(feat=128, dim=256, classes=3, k=10, nodes=100_000, bank=1_000_000, emulate_rank_of=0)
model, feats, adj, _ = bench.build_workload(args, dev, 0, 1, "keys")
labels = torch.randint(0, 3, (args.nodes,), device=dev)
model.train()
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
ms = BB.event_ms(lambda: BB._node_step_gpu(model, feats, adj, labels, opt), steps, warm=3)
model.eval()
with torch.no_grad():
    ms_inf = BB.event_ms(lambda: model(feats, adj), steps, warm=2)
print(f"fine-tuning step {ms:.3f} ms, inference forward {ms_inf:.3f} ms")
