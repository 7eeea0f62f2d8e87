import os, sys, types, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from ragraph_amd import kernels as K
from ragraph_amd.bank_build import build_reference_recipe_bank
from ragraph_amd.preprompt import PrePrompt
from ragraph_amd.data import synthetic_big_graph
from ragraph_amd.graph import CSRGraph
dev = torch.device("cuda:0")
torch.manual_seed(1)
F, D, C, N, n = 128, 256, 3, 1_000_000, 100_000
pre = PrePrompt(F, D, "prelu", 1, 0.3).to(dev)
adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, seed=8, device=dev), n)
feats = torch.randn(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
with torch.no_grad():
    pre.gcn.convs[0].bias.normal_(0, 0.1)
    tgb = build_reference_recipe_bank(pre, N, F, C, D, device=dev)
    h = pre.inference(feats, adj)
kn = tgb.keys_normalized
for B in (500, 4096, 100_000):
    q = h[:B].contiguous()
    s32, i32 = K.topk_cosine(q[:2000].contiguous(), kn, 10)
    kth = s32[:, 9]
    print(f"B={B}: exact k-th best of the first rows: min {float(kth.min()):.4f} median {float(kth.median()):.4f} max {float(kth.max()):.4f}")
    for spec in (False, True):
        index = K.KeyIndex(kn)
        index.spec_enabled = spec
        index.search_rows()
        inner = index.search_index
        inner.spec_enabled = spec
        for _ in range(5):
            s, i = index.topk(q, 10)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5 if B > 10000 else 30
        e0.record()
        for _ in range(reps):
            s, i = index.topk(q, 10)
        e1.record()
        torch.cuda.synchronize()
        st = inner._spec.get(10, {})
        print(f"   spec={spec}: {e0.elapsed_time(e1) / reps:.4f} ms  last_prior {inner.last_prior}  state: used {st.get('used')} failed {st.get('failed')} off_at {st.get('off_at')} "
              f"cand(bound) {st.get('cand')} hist {[(round(a, 3), round(b, 3)) for a, b in st.get('hist', [])][-3:]}  i8_off {inner._i8_off} overflowed {index.overflowed_queries}")
