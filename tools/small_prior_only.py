"""us per call of the single-launch kernel under a forced speculative first bound only (for libraries built with
-DRG_SMALL_FORCE_PRIOR, whose calls without a prior are invalid):  RAGRAPH_HIP_SO=... python tools/small_prior_only.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K
dev = torch.device("cuda", 0)
kn = K.normalize_rows(torch.randn(1_000_000, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
kb = K.keys_to_bf16(kn)
out = []
for B in (1, 4, 16, 32):
    q = torch.randn(B, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
    s32, i32 = K.topk_cosine(q, kn, 10)
    K.set_filter_prior(float(s32[:, 9].min()) - 0.02)
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
    out.append(f"B={B}: {sorted(rounds)[1] * 1e3:.1f} us")
print(os.environ.get("RAGRAPH_HIP_SO", "product"), "  ".join(out), flush=True)
