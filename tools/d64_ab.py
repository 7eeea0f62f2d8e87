#!/usr/bin/env python3
"""The edge flavour's retrieval calls of up to 256 queries (D = 64, 4M-key bank): ms per call through KeyIndex, for an A/B of
the direct kernel on the int8 copy (RAGRAPH_FILTER_I8_DIRECT_D64=0/1, read once per process):
    RAGRAPH_FILTER_I8_DIRECT_D64=0 python tools/d64_ab.py [N] [spec]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
spec = len(sys.argv) > 2 and sys.argv[2] == "spec"
dev = torch.device("cuda", 0)
kn = K.normalize_rows(torch.randn(N, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
index = K.KeyIndex(kn)
index.spec_enabled = spec
os.environ["RAGRAPH_TOPK_SMALL"] = "0"   # (the multi-launch path at every size)
out = []
for B in (1, 17, 64, 128, 256):
    q = torch.randn(B, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
    for _ in range(5):
        s, i = index.topk(q, 10)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        s, i = index.topk(q, 10)
    e1.record()
    torch.cuda.synchronize()
    s32, i32 = K.topk_cosine(q, kn, 10)
    out.append(f"B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us ({'same bits' if torch.equal(i, i32) and torch.equal(s, s32) else 'DIFFERENT'}, "
               f"i8 levels {K.filtered_i8_levels(B, N, 64, 10)})")
print(f"D64 int8 direct = {os.environ.get('RAGRAPH_FILTER_I8_DIRECT_D64', '1')}, spec = {spec}, bank {N} x 64 |", "  ".join(out),
      "| overflowed", index.overflowed_queries, flush=True)
