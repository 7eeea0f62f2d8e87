"""Is the bf16 filter kernel limited by its schedule or by the clock the chip holds?  The same launch (no key passes:
RAGRAPH_FILTER_ABLATE=1) over a random bank and over a constant bank: identical instruction streams, different switching
activity.  MI355X: 36.3 ms vs 27.9 ms."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K
dev = torch.device("cuda:0")
B, N, D, k = 100000, 1000000, 256, 10
for name in ("random", "constant"):
    g = torch.Generator(device=dev).manual_seed(0)
    if name == "random":
        kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
        q = torch.randn(B, D, device=dev, generator=g)
    else:
        kn = K.normalize_rows(torch.ones(N, D, device=dev))
        q = torch.ones(B, D, device=dev)
    kb = K.keys_to_bf16(kn)
    L = K.N.lib(); prof = L.ragraph_filter_profile_create(); L.ragraph_filter_profile_attach(prof)
    for _ in range(3):
        K.topk_cosine_filtered(q, kn, kb, k)
        ms = L.ragraph_filter_profile_last_ms(prof)
    print(name, "filter kernel ms:", round(ms, 2), flush=True)
    del kn, q, kb
