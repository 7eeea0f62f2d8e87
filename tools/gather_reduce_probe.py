#!/usr/bin/env python3
"""ragraph_gather_reduce_f32 at c2's shape (100 000 queries x k = 10 winners of a 1M x 256 value table + 3 label columns)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
N, D, C, B, k = 1_000_000, 256, 3, 100_000, 10
V = torch.randn(N, D, device=dev, generator=g)
L = torch.randn(N, C, device=dev, generator=g)
idx = torch.randint(0, N, (B, k), device=dev, generator=g)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = t(lambda: K.gather_reduce(V, L, idx))
print(f"gather_reduce B={B} k={k} D={D}: {us:.1f} us = {B * k * D * 4 / us / 1e6:.2f} TB/s of row gathers")
V64 = torch.randn(4_000_000, 64, device=dev, generator=g)
idx2 = torch.randint(0, 4_000_000, (B, k), device=dev, generator=g)
us = t(lambda: K.gather_reduce(V64, None, idx2, v_scale=0.1))
print(f"gather_reduce B={B} k={k} D=64 (edge flavour): {us:.1f} us")
A = torch.randn(B, D, device=dev, generator=g)
two = t(lambda: K.axpby(A, 0.5, K.gather_reduce(V, L, idx)[0], 0.5))
one = t(lambda: K.gather_reduce_mix(V, L, idx, A, 0.5, 0.5))
print(f"reduce + axpby: {two:.1f} us; gather_reduce_mix: {one:.1f} us")
