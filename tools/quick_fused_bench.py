"""Small-bank retrieval: the single-launch kernel against the multi-launch dispatch it replaces (ms per call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K

dev = torch.device("cuda:0")
torch.manual_seed(0)


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


shapes = [(2708, 10000, 128, 5), (512, 10000, 128, 5), (256, 10000, 128, 5), (64, 10000, 128, 5), (16, 1113, 256, 3),
          (545, 1113, 256, 3), (1024, 10000, 256, 10), (4096, 5000, 256, 10), (2708, 10000, 64, 5), (500, 20000, 64, 10),
          (8192, 10000, 128, 5), (300, 40000, 64, 10)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for B, Nk, D, k in shapes:
    kn = K.normalize_rows(torch.randn(Nk, D, device=dev))
    q = torch.randn(B, D, device=dev)
    kb = K.keys_to_bf16(kn)
    os.environ["RAGRAPH_TOPK_FUSED"] = "0"
    idx_old = K.KeyIndex(kn)
    t_old = t(lambda: idx_old.topk(q, k))
    os.environ["RAGRAPH_TOPK_FUSED"] = "1"
    if K.N.lib().ragraph_topk_cosine_fused_ok(B, Nk, D, k):
        t_new = t(lambda: K.topk_cosine_fused(q, kn, kb, k))
        s1, i1 = idx_old.topk(q, k) if False else K.topk_cosine(q, kn, k)
        s2, i2 = K.topk_cosine_fused(q, kn, kb, k)
        same = bool(torch.equal(i1, i2) and torch.equal(s1, s2))
    else:
        t_new, same = float("nan"), None
    print(f"{B:6d} x {Nk:6d} x {D:3d} k={k:2d}: dispatch without fused {t_old:.4f} ms, fused {t_new:.4f} ms, same bits {same}, "
          f"copy {2 * Nk * D / 2**20:.1f} MiB, helps={K.fused_helps(B, Nk, D, k)}", flush=True)
