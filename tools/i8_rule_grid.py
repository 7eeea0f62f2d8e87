"""ms per call over a grid of mid-size shapes under the CURRENT rule for int8 levels (the schedule's cost model; its int8 candidate
factor is read once per process from RAGRAPH_FILTER_I8_CANDF): the filtered call with its bound pass, and KeyIndex in its steady
state (speculative first bound).  Run once per setting and compare the tables:
    RAGRAPH_FILTER_I8_CANDF=2.0 python tools/i8_rule_grid.py [D ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ragraph_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)
dims = [int(a) for a in sys.argv[1:]] or [256, 128, 64]
k = 10


def ms(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("candf", os.environ.get("RAGRAPH_FILTER_I8_CANDF", "default"), flush=True)
for D in dims:
    for N in (70000, 150000, 300000, 500000, 1000000):
        g = torch.Generator(device=dev).manual_seed(N + D)
        kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
        kb = K.keys_to_bf16(kn)
        index = K.KeyIndex(kn, dedup=False)
        for B in (300, 512, 1100, 2048, 4096, 8192, 16384):
            q = torch.randn(B, D, device=dev, generator=g)
            t_f = ms(lambda: K.topk_cosine_filtered(q, kn, kb, k))
            for _ in range(4):   # (the index learns its prior from the first calls' statistics)
                index.topk(q, k)
                torch.cuda.synchronize()
            t_i = ms(lambda: index.topk(q, k))
            print(f"D={D:3d} N={N:7d} B={B:5d} i8={K.filtered_i8_levels(B, N, D, k)}  filtered {t_f:.4f}  index {t_i:.4f}"
                  f"{'  (prior)' if index.last_prior is not None else ''}", flush=True)
