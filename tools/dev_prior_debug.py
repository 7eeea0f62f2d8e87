import os, sys, types
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
args = types.SimpleNamespace(feat=128, dim=256, classes=3, k=10, nodes=100_000, bank=1_000_000, emulate_rank_of=G, key_shards=2)
model, feats, adj, _ = bench.build_workload(args, dev, 0, 1, "keys")
tgb = model.toy_graph_base
orig = tgb.prior.record
def rec(k, spec, misses, lo, hi, cand, over):
    ok = orig(k, spec, misses, lo, hi, cand, over)
    print(f"call {tgb.prior.calls}: spec={spec} misses={misses} lo={lo:.4f} hi={hi:.4f} cand={cand:.1f} over={over} -> ok={ok} state={ {kk: (v if kk != 'hist' else len(v)) for kk, v in tgb.prior._st[k].items()} }")
    return ok
tgb.prior.record = rec
for s in range(8):
    f = torch.randn(args.nodes, args.feat, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + s))
    with torch.no_grad():
        model(f, adj)
    torch.cuda.synchronize()
