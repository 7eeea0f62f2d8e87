#!/usr/bin/env python3
"""Phase stamps of the single-launch kernel (a -DRG_SMALL_TIMING build: RAGRAPH_HIP_SO=build_ab/lib_smalltiming.so).
    RAGRAPH_HIP_SO=... python tools/small_timing.py [B ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
N, D, k = 1_000_000, 256, 10
kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
kb = K.keys_to_bf16(kn)
for B in [int(a) for a in sys.argv[1:]] or [1, 16]:
    q = torch.randn(B, D, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
    for rep in range(3):
        print(f"B = {B} call {rep}", file=sys.stderr, flush=True)
        prior = os.environ.get("SMALL_TIMING_PRIOR")   # (a forced speculative first bound: the product's steady state)
        if prior:
            K.set_filter_prior(float(prior))
        try:
            K.topk_cosine_small(q, kn, kb, k)
        finally:
            K.set_filter_prior(None)
        torch.cuda.synchronize()
