"""Retrieval throughput of the product dispatch (KeyIndex.topk) against batch size, with the exact fp32 kernels beside
it: SURVEY.md §8(d)'s B list.  Development aid; bench.py is the judged harness.

  python tools/batch_sweep.py [N [D [k]]]        # default 1000000 256 10
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ragraph_amd import kernels as K

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
BS = [int(x) for x in os.environ.get("SWEEP_B", "1,4,16,32,48,64,128,129,256,512,1024,2048,4096,12500,25000,50000,100000").split(",")]

dev = torch.device("cuda:0")
torch.manual_seed(0)
kn = K.normalize_rows(torch.randn(N, D, device=dev))
index = K.KeyIndex(kn)
HBM, F32, BF16 = 8.0e12, 157.3e12, 2516.6e12


def timed(fn, B):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    reps = 3 if B * N > 2e10 else (10 if B * N > 1e9 else 30)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"bank {N} x {D}, k = {k}")
print(f"{'B':>7} {'path':>9} {'ms':>9} {'q/s':>10} {'bank GB/s':>10} {'TFLOP/s':>8}   {'fp32 ms':>9} {'speed-up':>8}  bound (fraction)")
for B in BS:
    q = torch.randn(B, D, device=dev)
    if os.environ.get("SWEEP_FORCE_FILTER"):  # crossover hunting: the filtered path whatever filter_helps says
        K.filter_helps = lambda *a, **kw: True
    filtered = K.filter_helps(B, N, D, k)
    ms = timed(lambda: index.topk(q, k), B)
    if K.packed_keys_help(B, D, k) and index._packed is None:
        index._packed = K.pack_keys(kn)  # the fp32 comparison below runs on its best copy
    kp = index._packed if K.packed_keys_help(B, D, k) else None
    if filtered:
        ms32 = timed(lambda: K.topk_cosine(q, kn, k, keys_packed=kp), B) if B <= 25000 else float("nan")
    else:
        ms32 = ms
    flops, bank = 2.0 * B * N * D, 4.0 * N * D
    t = ms * 1e-3
    # which roofline binds this batch: one pass over the bank, or the score matrix on the MFMA pipe that computes it
    hbm_t, mfma_t = bank / HBM, flops / (BF16 if filtered else F32)
    bound = ("hbm %.2f" % (hbm_t / t)) if hbm_t > mfma_t else ("%s mfma %.2f" % ("bf16" if filtered else "fp32", mfma_t / t))
    print(f"{B:>7} {'filtered' if filtered else 'fp32':>9} {ms:9.3f} {B / t:10.0f} {bank / t / 1e9:10.0f} {flops / t / 1e12:8.1f}   "
          f"{ms32:9.3f} {ms32 / ms:8.2f}  {bound}", flush=True)
