#!/usr/bin/env python3
"""A/B of the single-launch kernel for a handful of queries (csrc/topk_small.hip) against the multi-launch filtered call:
ms per call through KeyIndex at B = 1 .. 32 on an N x D bank (RAGRAPH_TOPK_SMALL=0/1, read per call).
    python tools/small_ab.py [N] [D] [k]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
index = K.KeyIndex(kn)
K.SMALL_MAX_B = 32    # (the kernel takes up to 32 queries; the product dispatch stops where the filtered call wins)
for _ in range(6):    # (warm the index's statistics: both paths then run under the bank's speculative first bound)
    index.topk(torch.randn(256, D, device=dev), k)
    torch.cuda.synchronize()
print(f"bank {N} x {D}, k = {k}; ms per call (20 reps after 5 warm-ups)")
for B in (1, 2, 4, 8, 16, 24, 32):
    q = torch.randn(B, D, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
    row = []
    outs = []
    for small in ("0", "1"):
        os.environ["RAGRAPH_TOPK_SMALL"] = small
        for _ in range(5):
            s, i = index.topk(q, k)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            s, i = index.topk(q, k)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 20)
        outs.append((s, i))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    print(f"B = {B:3d}: filtered (multi-launch) {row[0]:.4f}   one launch {row[1]:.4f}   same bits: {same}   prior {index.search_index.last_prior}", flush=True)
