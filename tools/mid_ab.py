#!/usr/bin/env python3
"""ms per retrieval call through KeyIndex at mid batch sizes (default 64 128 256 512 1024 4096) on an N x D bank, k = 10 --
for A/Bs of schedule switches that are read once per process (RAGRAPH_FILTER_FORCE_N0 / _L, RAGRAPH_FILTER_I8_DIRECT ...):
    RAGRAPH_FILTER_FORCE_N0=65536 RAGRAPH_FILTER_FORCE_L=1 python tools/mid_ab.py 256 512"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K  # noqa: E402

Bs = [int(a) for a in sys.argv[1:]] or [64, 128, 256, 512, 1024, 4096]
N, D, k = int(os.environ.get("MID_N", 1_000_000)), int(os.environ.get("MID_D", 256)), int(os.environ.get("MID_K", 10))
dev = torch.device("cuda", 0)
kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
index = K.KeyIndex(kn)
tag = " ".join(f"{k_}={v}" for k_, v in sorted(os.environ.items()) if k_.startswith("RAGRAPH_"))
out = []
for B in Bs:
    q = torch.randn(B, D, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
    for _ in range(5):
        index.topk(q, k)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            index.topk(q, k)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    out.append(f"B{B} {best:.4f}")
print(f"[{tag or 'product'}] " + "  ".join(out), flush=True)
