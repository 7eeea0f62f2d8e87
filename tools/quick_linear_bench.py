"""Ad-hoc timing of ragraph_linear_f32 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K

dev = torch.device("cuda:0")
shapes = [(100000, 128, 256), (100000, 256, 256), (100000, 256, 128), (4096, 256, 1000000 // 8)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for M, Kd, N in shapes:
    X = torch.randn(M, Kd, device=dev)
    W = torch.randn(N, Kd, device=dev)
    b = torch.randn(N, device=dev)
    for _ in range(3):
        K.linear(X, W, b, act=2, alpha=0.25)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.linear(X, W, b, act=2, alpha=0.25)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    by = 4.0 * (M * Kd + N * Kd + M * N)
    print(f"linear M={M} K={Kd} N={N}: {ms*1e3:.1f} us  {2.0*M*Kd*N/ms/1e9:.1f} TFLOP/s  {by/ms/1e6:.0f} GB/s", flush=True)
