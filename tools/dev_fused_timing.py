import os, sys
sys.path.insert(0, "/root/repo")
import torch
from ragraph_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, Nk, D, k in [(2708, 10000, 128, 5), (64, 10000, 128, 5)]:
    kn = K.normalize_rows(torch.randn(Nk, D, device=dev)); q = torch.randn(B, D, device=dev); kb = K.keys_to_bf16(kn)
    for _ in range(4):
        K.topk_cosine_fused(q, kn, kb, k)
    torch.cuda.synchronize()
    print("----", B, flush=True)
