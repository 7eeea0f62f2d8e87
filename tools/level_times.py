"""Per-level filter launch durations (HIP events inside the library) of one retrieval call, for schedule / ablation A/Bs:
python tools/level_times.py B [N D k reps]   (RAGRAPH_FILTER_ABLATE=1: no key passes -- the bare MFMA stream)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragraph_amd import kernels as K
from ragraph_amd import _native

B = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 256
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
kn = K.normalize_rows(torch.randn(N, D, device=dev))
index = K.KeyIndex(kn)
q = torch.randn(B, D, device=dev)
L = _native.lib()
for _ in range(2):
    index.topk(q, k)
torch.cuda.synchronize()
prof = L.ragraph_filter_profile_create(); L.ragraph_filter_profile_attach(prof)
acc = None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for _ in range(reps):
    e0.record()
    index.topk(q, k)
    e1.record()
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
    a_ms = (ctypes.c_float * 4)()
    a_i8 = (ctypes.c_int * 4)()
    a_keys = (ctypes.c_int64 * 4)()
    n = L.ragraph_filter_profile_levels(prof, a_ms, a_i8, a_keys)
    row = [a_ms[i] for i in range(4)]
    acc = row if acc is None else [x + y for x, y in zip(acc, row)]
L.ragraph_filter_profile_attach(None)
print(f"B={B} N={N} D={D} k={k}: call {tot / reps:.3f} ms; levels (ms, int8, keys): "
      + ", ".join(f"({acc[i] / reps:.3f}, {a_i8[i]}, {a_keys[i]})" for i in range(4)))
