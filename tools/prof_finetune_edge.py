"""The edge flavour's fine-tuning step at c5's shape (modules/RAGraph.py:335-355: edge dropout, forward through gate + 3
propagation layers + retrieval, BPR + L2, backward, Adam), repeated: run under rocprofv3 --kernel-trace --stats for the
per-kernel split of a step (round 6: no rocprim:: / hipcub:: kernel may appear -- the per-step graph rebuild runs on the
library's own radix sort and prefix sums).
    python tools/prof_finetune_edge.py [steps] [device|host]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tools"))
import torch

import bench_blocks as BB

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
_, m5 = BB.config_c5(dev, False)
if len(sys.argv) > 2:
    m5.dropout_rng = sys.argv[2]
U, I = m5.num_users, m5.num_items
g = torch.Generator().manual_seed(76)
batch = (torch.randint(0, U, (4096,), generator=g), torch.randint(0, I, (4096,), generator=g), torch.randint(0, I, (4096,), generator=g))
params = [p for p in m5.parameters() if p.requires_grad]
m5.train()
opt = torch.optim.Adam(params, lr=1e-3)


def step():
    opt.zero_grad()
    loss, _ = m5.cal_loss(batch)
    loss.backward()
    opt.step()


ms = BB.event_ms(step, steps, warm=1)
print(f"edge fine-tuning step {ms:.1f} ms (dropout mask drawn on the {m5.dropout_rng})")
