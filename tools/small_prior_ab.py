import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K
dev = torch.device("cuda", 0)
kn = K.normalize_rows(torch.randn(1_000_000, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
kb = K.keys_to_bf16(kn)
for B in (1, 4, 16, 32):
    q = torch.randn(B, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
    s32, i32 = K.topk_cosine(q, kn, 10)
    lo = float(s32[:, 9].min())
    res = []
    for prior in (None, lo - 0.02):
        K.set_filter_prior(prior)
        for _ in range(10):
            K.topk_cosine_small(q, kn, kb, 10)
        torch.cuda.synchronize()
        rounds = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                s, i, over = K.topk_cosine_small(q, kn, kb, 10)
            e1.record()
            torch.cuda.synchronize()
            rounds.append(e0.elapsed_time(e1) / 200)
        K.set_filter_prior(None)
        assert torch.equal(i, i32) and torch.equal(s, s32) and int(over) == 0
        res.append(sorted(rounds)[1] * 1e3)
    print(f"B={B}: bound phase {res[0]:.1f} us   prior {res[1]:.1f} us", flush=True)
