"""The int8 copy's two classes on a bank: the classes, and candidates per query / time of a filtered call.
  python tools/i8_classes_probe.py [gauss|clustered|onehot] [B] [N] [D]     (RAGRAPH_I8_ONE_SCALE=1: the single scale)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K

kind = sys.argv[1] if len(sys.argv) > 1 else "gauss"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
D = int(sys.argv[4]) if len(sys.argv) > 4 else 256
k = 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(31)
if kind == "clustered":
    c = torch.randn(40, D, device=dev, generator=g)
    kn = K.normalize_rows(c[torch.randint(0, 40, (N,), device=dev, generator=g)] + 1.2 * torch.randn(N, D, device=dev, generator=g))
    q = c[torch.randint(0, 40, (B,), device=dev, generator=g)] * (torch.rand(B, 1, device=dev, generator=g) * 2.0) + \
        torch.randn(B, D, device=dev, generator=g)
else:
    kn = torch.randn(N, D, device=dev, generator=g)
    if kind == "onehot":
        kn[N // 3] = 0
        kn[N // 3, 7] = 1
    kn = K.normalize_rows(kn)
    q = torch.randn(B, D, device=dev, generator=g)
kb = K.keys_to_bf16(kn)
print(kind, B, N, D, "one-scale" if os.environ.get("RAGRAPH_I8_ONE_SCALE") == "1" else "two scales", K.int8_copy_classes(kb, N), flush=True)
print("int8 levels planned:", K.filtered_i8_levels(B, N, D, k))
s, i, over, st = K.topk_cosine_filtered(q, kn, kb, k, return_stats=True)
torch.cuda.synchronize()
print("levels (dtype, keys, candidates/query):", K.filter_stats_levels(st.cpu().tolist()), "overflow", int(over))
s0, i0 = K.topk_cosine(q, kn, k)
print("exact:", bool(torch.equal(i0, i) and torch.equal(s0, s)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    K.topk_cosine_filtered(q, kn, kb, k)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    K.topk_cosine_filtered(q, kn, kb, k)
e1.record()
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 10:.4f} ms per call")
