"""One batch size of the product retrieval dispatch, repeated: run under rocprofv3 --kernel-trace --stats for the
per-kernel breakdown of a small-batch call (tools/gpu_small_batch.sh).   python tools/prof_small_batch.py B [N D k reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragraph_amd import kernels as K

B = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 256
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
dev = torch.device("cuda:0")
torch.manual_seed(0)
kn = K.normalize_rows(torch.randn(N, D, device=dev))
index = K.KeyIndex(kn)
q = torch.randn(B, D, device=dev)
if os.environ.get("RAGRAPH_NO_PRIOR") == "1":   # (the calls with their bound pass: A/B against the speculative first bound)
    index.spec_enabled = False
for _ in range(5):      # (the dispatch settles: the calls' statistics -- overflow, the speculative bound -- arrive one call late)
    index.topk(q, k)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    index.topk(q, k)
e1.record()
torch.cuda.synchronize()
print(f"B={B} N={N} D={D} k={k}: {e0.elapsed_time(e1) / reps:.4f} ms per call"
      f" (speculative first bound: {index.search_index.last_prior})")
