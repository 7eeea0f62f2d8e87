import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K
dev = torch.device("cuda:0")
torch.manual_seed(0)
kn = K.normalize_rows(torch.randn(1_000_000, 256, device=dev)); index = K.KeyIndex(kn)
for B in (256, 64):
    q = torch.randn(B, 256, device=dev)
    for _ in range(3):
        index.topk(q, 10)
    torch.cuda.synchronize()
    print("----", B, file=sys.stderr, flush=True)
