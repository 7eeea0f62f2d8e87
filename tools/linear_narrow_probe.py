#!/usr/bin/env python3
"""TaskDecoder's fc2 at c2 (100 000 x 256 -> 3): RAGRAPH_LINEAR_NARROW=0|1 python tools/linear_narrow_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K
dev = torch.device("cuda", 0)
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, Kd, N in ((100_000, 256, 3), (100_000, 256, 7), (12_500, 256, 3), (4_000_000, 64, 2)):
    X = torch.randn(M, Kd, device=dev); W = torch.randn(N, Kd, device=dev) * 0.1; b = torch.randn(N, device=dev)
    print(f"RAGRAPH_LINEAR_NARROW={os.environ.get('RAGRAPH_LINEAR_NARROW', '1')}: {M} x {Kd} -> {N}: {t(lambda: K.linear(X, W, b)):.1f} us")
