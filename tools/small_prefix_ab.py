import os, sys, torch
sys.path.insert(0, "/root/repo")
from ragraph_amd import kernels as K
dev = torch.device("cuda", 0)
N, D, k = 1_000_000, 256, 10
kn = K.normalize_rows(torch.randn(N, D, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)))
index = K.KeyIndex(kn)
for base in [int(a) for a in os.environ.get('BASES', '16384,12288,20480,24576,16384,12288').split(',')]:
    os.environ["RAGRAPH_SMALL_PREFIX_BASE"] = str(base)
    row = []
    for B in [int(a) for a in os.environ.get('BS', '2,4,8').split(',')]:
        q = torch.randn(B, D, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + B))
        for _ in range(5):
            s, i = index.topk(q, k)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                s, i = index.topk(q, k)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        row.append(f"B{B} {best:.4f}")
    print(f"prefix base {base}: " + "  ".join(row) + f"  overflowed {index.overflowed_queries}", flush=True)
