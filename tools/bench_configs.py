#!/usr/bin/env python3
"""Timings of the other BASELINE.json configs (parity-test shapes, not the judged bench line):
  c1  RAGraph_node, Cora-shaped graph (2708 nodes, F=1433), 10k x 128 bank, k=5      -- eager and HIP-graph replay
  c3  RAGraph_graph, PROTEINS-style batches of 16 graphs, 1113-key bank, k=3          -- batched forward, eager / replay
  c5  RAGraph_edge generate(): 3-layer propagation + retrieval of all nodes vs a 4M x 64 bank (k=10)
One JSON object per line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ragraph_amd import kernels as K
from ragraph_amd.data import DataLoader, synthetic_bipartite, synthetic_big_graph, synthetic_tu_dataset
from ragraph_amd.graph import CSRGraph
from ragraph_amd.preprompt import PrePrompt
from ragraph_amd.RAGraph import RAGraph, RAGraphGraph

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def bank(model, N, D, C):
    model.toy_graph_base.add_resources(torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=-1),
                                       torch.randn(N, D, device=dev),
                                       torch.nn.functional.one_hot(torch.randint(0, C, (N,), device=dev), C).float())
    _ = model.toy_graph_base.keys_normalized


with torch.no_grad():
    # ---- c1 --------------------------------------------------------------------------------------------------
    n, F, D, C = 2708, 1433, 128, 7
    adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 4, seed=7, device=dev), n)
    X = (torch.rand(n, F, device=dev) < 0.0127).float()
    X = X / X.sum(1, keepdim=True).clamp_min(1)
    m1 = RAGraph(PrePrompt(F, D, "prelu", 1, 0.3).to(dev), None, F, C, D, device=dev).eval()
    m1.toy_graph_base.retrieve_num = 5
    bank(m1, 10_000, D, C)
    _ = adj.row_normalized_values()
    f1 = lambda: m1(X, adj)
    te, tg = timeit(f1), timeit(capture(f1))
    print(json.dumps({"config": "c1 RAGraph_node Cora-shaped 2708 nodes, 10k x 128 bank, k=5", "eager_ms": round(te * 1e3, 3),
                      "hipgraph_replay_ms": round(tg * 1e3, 3), "nodes_per_s_replay": round(n / tg)}), flush=True)
    # ---- c3 --------------------------------------------------------------------------------------------------
    ds = synthetic_tu_dataset(num_graphs=1113, num_node_attributes=1, num_node_labels=3, num_classes=2, seed=9)
    F3 = 4  # 1 attribute + 3 one-hot node labels are all fed as features here (F must be a kernel-friendly width)
    m3 = RAGraphGraph(PrePrompt(F3, 256, "prelu", 1, 0.3).to(dev), None, F3, 2, 256, device=dev).eval()
    bank(m3, 1113, 256, 2)
    b = next(iter(DataLoader(ds, batch_size=16)))
    Xb = b.x.to(dev)
    ab = CSRGraph.from_edge_index_sym_normalized(b.edge_index.to(dev), Xb.shape[0])
    ptr = b.ptr.to(dev)
    _ = ab.row_normalized_values()
    f3 = lambda: m3.forward_batch(Xb, ab, ptr)
    te, tg = timeit(f3), timeit(capture(f3))
    print(json.dumps({"config": "c3 RAGraph_graph 16 PROTEINS-style graphs per pass (%d nodes), 1113-key bank, k=3" % Xb.shape[0],
                      "eager_ms": round(te * 1e3, 3), "hipgraph_replay_ms": round(tg * 1e3, 3),
                      "graphs_per_s_replay": round(16 / tg)}), flush=True)
    # ---- few-shot node flavour: structural + semantic retrieval on a batch of 16 graphs (n ~ 528) ------------------
    from ragraph_amd.RAGraph_fewshot import RAGraph as RAGraphFewShot, _dense
    bf = next(iter(DataLoader(synthetic_tu_dataset(num_graphs=64, num_node_attributes=18, num_node_labels=3, seed=11), batch_size=16)))
    Xf = torch.rand(bf.x.shape[0], 18, device=dev)
    af = CSRGraph.from_edge_index_sym_normalized(bf.edge_index.to(dev), Xf.shape[0])
    logits = torch.randn(3, 256, device=dev)
    mf = RAGraphFewShot(PrePrompt(18, 256, "prelu", 2, 0.3).to(dev), None, logits, 256, device=dev, dataset_name="ENZYMES").eval()
    Nf = 20_000
    mf.toy_graph_base.add_resources(torch.nn.functional.normalize(torch.randn(Nf, 256, device=dev), dim=-1),
                                    torch.randn(Nf, 256, device=dev),
                                    torch.nn.functional.one_hot(torch.randint(0, 3, (Nf,), device=dev), 3).float(),
                                    torch.rand(Nf, 10, device=dev))
    anchors = torch.randint(0, Xf.shape[0], (10,), device=dev)
    _ = af.row_normalized_values()
    tf = timeit(lambda: mf(Xf, af, logits, anchors=anchors), reps=20)
    tc = timeit(lambda: K.position_codes_csr(af.rowptr, af.col, af.val, anchors, 10.0), reps=20)
    tfw = timeit(lambda: K.position_code(K.floyd_warshall(_dense(af)), anchors, 10.0), reps=5)   # round 2's per-forward path
    print(json.dumps({"config": "few-shot RAGraph_node forward, 16 graphs per pass (%d nodes), 20k x 256 bank + 10-d position "
                                "codes, k=5" % Xf.shape[0], "forward_ms": round(tf * 1e3, 3),
                      "position_codes_csr_ms": round(tc * 1e3, 3), "all_pairs_floyd_warshall_codes_ms": round(tfw * 1e3, 3),
                      "forward_ms_with_all_pairs_codes": round((tf - tc + tfw) * 1e3, 3)}), flush=True)
    # ---- c5 --------------------------------------------------------------------------------------------------
    from ragraph_amd.RAGraph_edge import RAGraph as RAGraphEdge
    U, I = 2_200_000, 1_800_000
    edges, norm, times = synthetic_bipartite(U, I, edges_per_user=10, seed=10, device=dev)

    class DSx:
        num_users, num_items = U, I
    DSx.edges, DSx.edge_norm, DSx.edge_times = edges, norm, times

    class Pre:
        def generate(self):
            g = torch.Generator(device=dev).manual_seed(3)
            return 0.1 * torch.randn(U, 64, device=dev, generator=g), 0.1 * torch.randn(I, 64, device=dev, generator=g)
    t0 = time.perf_counter()
    m5 = RAGraphEdge(DSx, Pre(), phase="finetune", use_RAG=True, retrieve_num=10, device=dev).eval()
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    t5 = timeit(lambda: m5.generate(), reps=2)
    m5.use_RAG = False
    t5p = timeit(lambda: m5.generate(), reps=2)   # time encoding + 3 propagation layers alone
    m5.use_RAG = True
    nn_ = U + I
    print(json.dumps({"config": "c5 RAGraph_edge generate(): %d nodes, %d directed edges, 4M x 64 bank, k=10" % (nn_, edges.shape[0]),
                      "bank_build_s": round(t_build, 3), "generate_ms": round(t5 * 1e3, 1), "propagation_only_ms": round(t5p * 1e3, 1),
                      "retrieved_queries_per_s": round(nn_ / t5), "retrieval_TFLOPs": round(2.0 * nn_ * nn_ * 64 / t5 / 1e12, 1)}),
          flush=True)
