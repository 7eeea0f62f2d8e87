"""Ad-hoc timing of the fused top-k kernel (development aid; bench.py is the judged harness)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(1, 1_000_000, 256, 10), (16, 1_000_000, 256, 10), (256, 1_000_000, 256, 10), (4096, 1_000_000, 256, 10),
          (32768, 1_000_000, 256, 10), (4096, 4_000_000, 64, 10)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for B, N, D, k in shapes:
    kn = K.normalize_rows(torch.randn(N, D, device=dev))
    q = torch.randn(B, D, device=dev)
    kp = K.pack_keys(kn) if (K.packed_keys_help(B, D, k) and not os.environ.get("QTB_NO_PACK")) else None
    filt = bool(os.environ.get("QTB_FILTER")) and D in (64, 128, 256)
    K.filter_helps = lambda *a, **kw: False  # the plain entry below is the fp32 path
    kb = K.keys_to_bf16(kn) if filt else None
    if filt:
        def run():
            return K.topk_cosine_filtered(q, kn, kb, k, keys_packed=kp)
    else:
        def run():
            return K.topk_cosine(q, kn, k, keys_packed=kp)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    reps = 3 if B * N > 1e10 else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * N * D
    by = 4.0 * N * D
    tag = ("bf16-filter ovf=%d" % out[2]) if filt else ('packed' if kp is not None else 'natural')
    print(f"[{tag}] B={B} N={N} D={D} k={k}: {ms:.3f} ms  {B / ms * 1e3:.0f} q/s  {fl / ms / 1e9:.1f} TFLOP/s  bank-pass {by / ms / 1e6:.0f} GB/s",
          flush=True)
    del kn, q, kp, kb
