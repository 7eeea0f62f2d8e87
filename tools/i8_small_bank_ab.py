#!/usr/bin/env python3
"""Banks below the int8 levels' cut-offs (32 768 keys at D = 256 from 2048 queries, 65 536 otherwise): ms per filtered call
with the levels forced onto bf16 (RAGRAPH_FILTER_I8=0), forced onto int8 (= the number of levels) and as the rule decides.
    python tools/i8_small_bank_ab.py [mid]      (mid: banks of 70 k - 500 k keys whose batches the rule keeps on bf16)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
L = K.N.lib()
MID = [(512, 150000, 256), (512, 300000, 256), (1100, 150000, 256), (1100, 300000, 256), (1100, 500000, 256), (2048, 70000, 256),
       (4096, 70000, 256), (700, 200000, 128), (2048, 200000, 128), (4096, 100000, 128), (2048, 300000, 64), (8192, 100000, 64)]
for B, N, D in MID if len(sys.argv) > 1 and sys.argv[1] == "mid" else [(2048, 20000, 256), (8192, 20000, 256), (16384, 20000, 256), (8192, 30000, 256), (1024, 40000, 256), (512, 60000, 256),
                (4096, 50000, 128), (16384, 50000, 128), (65536, 40000, 128), (8192, 40000, 64), (16384, 60000, 64), (65536, 50000, 64)]:
    g = torch.Generator(device=dev).manual_seed(B + N + D)
    kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
    kb = K.keys_to_bf16(kn)
    q = torch.randn(B, D, device=dev, generator=g)
    plan = (ctypes.c_int64 * 7)()
    nlev = L.ragraph_topk_cosine_filtered_plan(B, N, D, 10, plan)
    res = {}
    ref = None
    for name, env in (("bf16", "0"), ("int8", str(nlev)), ("rule", None)):
        if env is None:
            os.environ.pop("RAGRAPH_FILTER_I8", None)
        else:
            os.environ["RAGRAPH_FILTER_I8"] = env
        for _ in range(3):
            s, i, over = K.topk_cosine_filtered(q, kn, kb, 10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            s, i, over = K.topk_cosine_filtered(q, kn, kb, 10)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20
        if ref is None:
            ref = (s, i)
        assert torch.equal(ref[0], s) and torch.equal(ref[1], i) and int(over) == 0
    os.environ.pop("RAGRAPH_FILTER_I8", None)
    print(f"{B:6d} x {N:6d} x {D:3d} ({nlev} level(s), rule: {K.filtered_i8_levels(B, N, D, 10)} on int8): bf16 {res['bf16']:.4f}  int8 {res['int8']:.4f}  "
          f"rule {res['rule']:.4f} ms   int8 / bf16 = {res['int8'] / res['bf16']:.3f}", flush=True)
