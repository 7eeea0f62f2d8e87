"""One rank of tests/test_gpu_two_rank.py: TWO real processes driving the real HIP library through a process group.

Both ranks share cuda:0 (the test box has one GPU; RCCL refuses two ranks on one device, so the group is gloo and
ragraph_amd.sharded stages the collectives' device tensors through the host).  Each rank
  * builds the same seeded workload (a 200k x 256 bank, a 20k-node graph, one RAGraph_node model),
  * computes the single-process forward and top-k on the whole bank,
  * runs the KEY-sharded forward (RAGraph._forward_key_shard: its shard of the bank through
    ragraph_topk_cosine_filtered_sharded_f32 with the exchange callback issuing collectives between the call's
    launches, all_to_all of the lists, merge, query-sharded tail, all_gather of the outputs), the key-sharded
    ShardedToyGraphBase.topk (all_gather + merge) and the QUERY-sharded forward,
  * and reports whether each equals the single-process result bit for bit, plus the exchange phases it went through.
Started by tests/conftest.py BEFORE the pytest process touches the GPU.   usage: two_rank_worker.py RANK WORLD PORT OUT
"""
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

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        from ragraph_amd import kernels as K
        from ragraph_amd.data import synthetic_bank, synthetic_big_graph
        from ragraph_amd.graph import CSRGraph
        from ragraph_amd.preprompt import PrePrompt
        from ragraph_amd.RAGraph import RAGraph
        from ragraph_amd.sharded import QueryShard, ShardedToyGraphBase, shard_bounds

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        N, D, C, k, n, F = 200_000, 256, 3, 10, 20_000, 64
        torch.manual_seed(0)
        pre = PrePrompt(F, D, "prelu", 1, 0.3).to(dev)
        model = RAGraph(pre, None, F, C, D, finetune=True, device=dev).eval()
        model.toy_graph_base.retrieve_num = k
        adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 10, seed=8, device=dev), n)
        feats = torch.randn(n, F, device=dev, generator=torch.Generator(device=dev).manual_seed(4321))
        Kb, Vb, Lb = synthetic_bank(N, D, C, device=dev)
        Kb = K.normalize_rows(Kb)
        single = model.toy_graph_base
        single.set_resources(Kb, Vb, Lb)
        with torch.no_grad():
            want = model(feats, adj)
            h = pre.inference(feats, adj)
            want_s, want_i = single.topk(h, k)
        res["filtered_path"] = bool(K.filter_helps(n, N // world, D, k))

        lo, hi = shard_bounds(N, world, rank)
        sharded = ShardedToyGraphBase(Kb[lo:hi].contiguous(), Vb, Lb, lo, k, values_replicated=True)
        model.toy_graph_base = sharded
        with torch.no_grad():
            got = model(feats, adj)                      # -> _forward_key_shard
            got_s, got_i = sharded.topk(h, k)            # all_gather of the lists + merge of all rows
        res["key_shard_forward_equal"] = bool(torch.equal(got, want))
        res["key_shard_topk_equal"] = bool(torch.equal(got_i, want_i) and torch.equal(got_s, want_s))
        res["exchange_count"] = {str(p): c for p, c in sharded.exchange_count.items()}

        model.toy_graph_base = single
        model.query_shard = QueryShard()
        with torch.no_grad():
            got_q = model(feats, adj)                    # -> _forward_query_shard
        res["query_shard_forward_equal"] = bool(torch.equal(got_q, want))
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
