import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K
dev = torch.device("cuda", 0)
kn = K.normalize_rows(torch.randn(1_000_000, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
index = K.KeyIndex(kn)
for _ in range(6):
    index.topk(torch.randn(256, 256, device=dev), 10); torch.cuda.synchronize()
for B in (1, 16, 256):
    q = torch.randn(B, 256, device=dev)
    for _ in range(20):
        index.topk(q, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        index.topk(q, 10)
    t_host = (time.perf_counter() - t0) / 1000
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 1000
    print(f"B={B}: host enqueue {t_host * 1e6:.1f} us per call, wall incl. GPU {t_all * 1e6:.1f} us per call, prior {index.search_index.last_prior}")
q = torch.randn(1, 256, device=dev)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000):
    index.topk(q, 10)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
