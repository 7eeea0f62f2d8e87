"""The single-launch kernel with a tiny list cap (a -DRG_SMALL_LIST_CAP=64 build: RAGRAPH_HIP_SO=build_ab/lib_smallcap.so), so that
lists overflow on an ordinary bank: with its bound pass (the last workgroup scans) and under forced priors (the fixup launch's
sliced scan answers them, together with the prior's misses) -- bits against the fp32 kernel, the overflow counts, ms per call.
    RAGRAPH_HIP_SO=build_ab/lib_smallcap.so python tools/small_cap_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(9)
N, D, k = 300_000, 256, 10
kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=g))
kb = K.keys_to_bf16(kn)
for B in (1, 5, 16, 32):
    q = torch.randn(B, D, device=dev, generator=g)
    if B > 2:
        q[1] = 0.0
    s0, i0 = K.topk_cosine(q, kn, k)
    kth = s0[:, k - 1]
    lo = float(kth[kth > 0].min())
    for prior in (None, lo - 0.15, lo - 0.02, lo + 0.01):
        K.set_filter_prior(prior)
        try:
            s, i, over, st = K.topk_cosine_small(q, kn, kb, k, return_stats=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                K.topk_cosine_small(q, kn, kb, k)
            e1.record()
            torch.cuda.synchronize()
        finally:
            K.set_filter_prior(None)
        w = st.cpu().tolist()
        ok = torch.equal(i, i0) and torch.equal(s, s0)
        print(f"B={B} prior={'none' if prior is None else f'{prior:.3f}'}: exact={ok} overflow={int(over)} misses={w[17]} word20={w[20]} "
              f"{e0.elapsed_time(e1) / 3:.3f} ms", flush=True)
        assert ok and int(over) == w[20]
