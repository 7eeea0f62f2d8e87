#!/usr/bin/env python3
"""Latency of a handful of queries through KeyIndex (the single-launch kernel): ms per call at B = 1, 4, 16 on the 1M x 256
bank, three rounds of 200 calls each, the median round.  A/B of two libraries on one box:
    RAGRAPH_HIP_SO=build_ab/lib_x.so python tools/small_lat.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
kn = K.normalize_rows(torch.randn(1_000_000, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
index = K.KeyIndex(kn)
out = []
for B in (1, 4, 16):
    q = torch.randn(B, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
    for _ in range(10):
        index.topk(q, 10)
    torch.cuda.synchronize()
    rounds = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            index.topk(q, 10)
        e1.record()
        torch.cuda.synchronize()
        rounds.append(e0.elapsed_time(e1) / 200)
    out.append(f"B={B}: {sorted(rounds)[1] * 1e3:.1f} us")
print(os.environ.get("RAGRAPH_HIP_SO", "in-tree library"), "|", "  ".join(out), "| overflowed", index.overflowed_queries, flush=True)
