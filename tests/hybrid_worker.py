"""One rank of tests/test_gpu_two_rank.py::test_four_ranks_hybrid_layout: FOUR real processes on the real HIP library, as 2 query
groups x 2 key shards (ragraph_amd.sharded.HybridLayout), all on cuda:0 under a gloo group (device tensors staged through the
host).  Each rank builds the same seeded workload, computes the single-process forward, runs RAGraph._forward_hybrid and
reports whether the gathered [n, C] output equals the single-process one bit for bit.  Started by tests/conftest.py BEFORE the
pytest process touches the GPU.   usage: hybrid_worker.py RANK WORLD PORT OUT"""
import datetime
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    res = {"rank": rank, "ok": False}
    try:
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=600))
        from ragraph_amd import kernels as K
        from ragraph_amd.data import synthetic_bank, synthetic_big_graph
        from ragraph_amd.graph import CSRGraph
        from ragraph_amd.preprompt import PrePrompt
        from ragraph_amd.RAGraph import RAGraph
        from ragraph_amd.sharded import HybridLayout, ShardedToyGraphBase

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        N, D, C, k, n, F = 160_000, 256, 3, 10, 17_001, 64      # (a ragged batch: 17 001 rows over 2 x 2 ranks)
        torch.manual_seed(0)
        pre = PrePrompt(F, D, "prelu", 1, 0.3).to(dev)
        model = RAGraph(pre, None, F, C, D, finetune=True, device=dev).eval()
        model.toy_graph_base.retrieve_num = k
        adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, seed=8, device=dev), n)
        feats = torch.randn(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
        Kb, Vb, Lb = synthetic_bank(N, D, C, device=dev)
        Kb = K.normalize_rows(Kb)
        model.toy_graph_base.set_resources(Kb, Vb, Lb)
        with torch.no_grad():
            want = model(feats, adj)
        layout = HybridLayout(2)
        lo, hi = layout.key_rows(N)
        tgb = ShardedToyGraphBase(Kb[lo:hi].contiguous(), Vb, Lb, lo, k, group=layout.key_group, values_replicated=True)
        model.toy_graph_base = tgb
        model.query_shard = layout.query_shard()
        with torch.no_grad():
            got = model(feats, adj)                      # -> _forward_hybrid
        res["layout"] = layout.name
        res["q_s"] = [layout.q, layout.s]
        res["hybrid_forward_equal"] = bool(torch.equal(got, want))
        res["exchange_count"] = {str(p): c for p, c in tgb.exchange_count.items()}
        # round 6: a FORCED speculative first bound that exactly one row of the batch misses (the midpoint of the two lowest
        # k-th best scores): only the key group that answers that row repeats its retrieval -- the decision is the group's --,
        # and the gathered output is still the single-process forward on every rank
        with torch.no_grad():
            h = pre.inference(feats, adj)
        full_s, _ = K.KeyIndex(Kb).topk(h, k)
        two = torch.sort(full_s[:, k - 1]).values[:2]
        tgb.prior.forced = float(0.5 * (two[0] + two[1]))
        r0 = tgb.reruns
        with torch.no_grad():
            got2 = model(feats, adj)
        tgb.prior.forced = None
        res["forced_one_forward_equal"] = bool(torch.equal(got2, want))
        res["forced_one_reruns"] = int(tgb.reruns - r0)
        res["world"] = dist.get_world_size()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
        res["ok"] = True
    except BaseException:
        res["error"] = traceback.format_exc()
    with open(os.path.join(out, f"rank{rank}.json"), "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
