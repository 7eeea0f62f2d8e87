"""BASELINE.json configs that round 1 only timed, now asserted on the HIP path, and the add_noise branches of retrieve.

  c1  Cora-shaped RAGraph_node forward (2708 nodes x 1433 features, 10 k x 128 bank, k = 5, C = 7) against the oracle's
      whole-forward restatement: encoder output and top-k indices bit-exact, logits to expf rounding.
  c4  8 M x 256 bank on ONE GPU as 8 row shards (idx_base) + topk_merge: equal to the unsharded call bit for bit through
      the fp32 kernels and through the bf16-filtered path, and to the oracle on a sample of the queries.
  noise  node (ToyGraphBase.py:66,73-79), graph (RAGraph_graph/.../ToyGraphBase.py:84-85,131-134) and edge
      (modules/RAGraph.py:308-321) add_noise branches against goldens the reference produced under torch.manual_seed.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cref, pipeline

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ---------------------------------------------------------------------------------------------------------------------
def test_large_inference_forward_mixes_behind_the_gather(dev, monkeypatch):
    """A node forward of more than RAGraph.OVERLAP_MIN_NODES nodes (hops on the side stream, one-launch encoder, the winners'
    value sum mixed into the prompt behind the gather: ragraph_gather_reduce_mix_f32) gives the bits of the path that keeps the
    entries apart (RAGRAPH_GATHER_MIX=0, RAGRAPH_SPMM_LINEAR=0), and of the pieces called one by one."""
    from ragraph_amd import autograd as A
    from ragraph_amd import kernels as K
    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph
    from ragraph_amd.ragraph_utils import Propagation

    n, F, D, C, N, k = 20_011, 128, 256, 3, 30_000, 10
    torch.manual_seed(3)
    adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 8, seed=11, device=dev), n)
    g = torch.Generator(device=dev).manual_seed(71)
    X = torch.randn(n, F, device=dev, generator=g)
    model = RAGraph(PrePrompt(F, D, "prelu", 1, 0.3).to(dev), None, F, C, D, finetune=True, device=dev).eval()
    assert n >= model.OVERLAP_MIN_NODES
    tgb = model.toy_graph_base
    tgb.retrieve_num = k
    tgb.add_resources(torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=-1),
                      torch.randn(N, D, device=dev, generator=g),
                      torch.nn.functional.one_hot(torch.randint(0, C, (N,), device=dev, generator=g), C).float())
    with torch.no_grad():
        fused = model(X, adj)
        monkeypatch.setenv("RAGRAPH_GATHER_MIX", "0")
        monkeypatch.setenv("RAGRAPH_SPMM_LINEAR", "0")
        apart = model(X, adj)
        monkeypatch.delenv("RAGRAPH_GATHER_MIX")
        monkeypatch.delenv("RAGRAPH_SPMM_LINEAR")
        torch.cuda.synchronize()
        assert torch.equal(fused, apart)
        h = model.pretrain_model.inference(X, adj)
        qe = Propagation.aggregate_k_hop_features(adj, h, model.query_graph_hop)
        rag, lab, _ = tgb.retrieve_reduced(h)
        hidden = K.axpby(qe, 1 - model.retrieve_weight, rag, model.retrieve_weight)
        by_hand = A.softmax_mix(model.decoder(hidden), lab, model.label_weight)
    assert torch.equal(fused, by_hand)


# ---------------------------------------------------------------------------------------------------------------------
def test_c1_cora_shaped_forward_matches_oracle(dev):
    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph

    n, F, D, C, N, k = 2708, 1433, 128, 7, 10_000, 5
    torch.manual_seed(0)
    # 5429 undirected edges on 2708 nodes = mean degree 4 (ring + random extras), binary-sparse row-normalised features
    adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 4, seed=7, device=dev), n)
    g = torch.Generator(device=dev).manual_seed(70)
    X = (torch.rand(n, F, device=dev, generator=g) < 0.0127).float()
    X = X / X.sum(1, keepdim=True).clamp_min(1)
    pre = PrePrompt(F, D, "prelu", 1, 0.3).to(dev)
    model = RAGraph(pre, None, F, C, D, finetune=True, device=dev).eval()
    model.toy_graph_base.retrieve_num = k
    keys = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=-1)
    vals = torch.randn(N, D, device=dev, generator=g)
    labs = torch.nn.functional.one_hot(torch.randint(0, C, (N,), device=dev, generator=g), C).float()
    model.toy_graph_base.add_resources(keys, vals, labs)
    with torch.no_grad():
        pre.gcn.convs[0].bias.normal_(0, 0.05)
        logits = model(X, adj)
        h = pre.inference(X, adj)
        s, idx = model.toy_graph_base.topk(h, k)
    conv, dec = pre.gcn.convs[0], model.decoder
    p = {"W": conv.fc.weight, "bias": conv.bias, "alpha": conv.act.weight, "fc1_w": dec.fc1.weight,
         "fc1_b": dec.fc1.bias, "fc2_w": dec.fc2.weight, "fc2_b": dec.fc2.bias}
    p = {kk: v.detach().cpu().numpy() for kk, v in p.items()}
    p["alpha"] = float(p["alpha"][0])
    csr = (adj.rowptr.cpu().numpy(), adj.col.cpu().numpy(), adj.val.cpu().numpy())
    ol, oi, oh = pipeline.node_forward(X.cpu().numpy(), csr, p, keys.cpu().numpy(), vals.cpu().numpy(),
                                       labs.cpu().numpy(), k, 3, 0.5, 0.5)
    assert np.array_equal(h.cpu().numpy(), oh)
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.allclose(logits.cpu().numpy(), ol, atol=1e-6)
    assert logits.shape == (n, C) and np.allclose(logits.sum(1).cpu().numpy(), 1.0, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------------
def test_c4_8m_bank_as_eight_shards(dev):
    from ragraph_amd import kernels as K
    from ragraph_amd.sharded import shard_bounds

    N, D, k, G = 8_000_000, 256, 10, 8
    g = torch.Generator(device=dev).manual_seed(44)
    kn = torch.empty(N, D, device=dev)
    for lo in range(0, N, 1_000_000):  # generated and normalised per shard: no second 8 GB temporary
        kn[lo:lo + 1_000_000] = K.normalize_rows(torch.randn(1_000_000, D, device=dev, generator=g))
    q = torch.randn(2048, D, device=dev, generator=g)
    q[0] = kn[N - 1] + 0.01 * q[0]      # winners in the last shard's last row ...
    q[1] = kn[5 * 1_000_000]            # ... and exactly on a shard boundary
    # fp32 kernels: unsharded vs 8 shards + merge
    full_s, full_i = K.topk_cosine(q, kn, k)
    ss, ii = [], []
    for r in range(G):
        lo, hi = shard_bounds(N, G, r)
        s, i = K.topk_cosine(q, kn[lo:hi], k, idx_base=lo)
        ss.append(s)
        ii.append(i)
    ms, mi = K.topk_merge(torch.stack(ss), torch.stack(ii))
    assert torch.equal(mi, full_i) and torch.equal(ms, full_s)
    assert int(full_i[0, 0]) == N - 1 and int(full_i[1, 0]) == 5_000_000
    # bf16-filtered path: per-shard (each rank's call at c4) and over the whole 8 M bank
    fs, fi = [], []
    for r in range(G):
        lo, hi = shard_bounds(N, G, r)
        shard = kn[lo:hi]
        assert K.filter_helps(q.shape[0], hi - lo, D, k)
        s, i, over = K.topk_cosine_filtered(q, shard, K.keys_to_bf16(shard), k, idx_base=lo)
        assert over == 0
        fs.append(s)
        fi.append(i)
    ms2, mi2 = K.topk_merge(torch.stack(fs), torch.stack(fi))
    assert torch.equal(mi2, full_i) and torch.equal(ms2, full_s)
    s8, i8, over = K.topk_cosine_filtered(q, kn, K.keys_to_bf16(kn), k)
    assert over == 0 and torch.equal(i8, full_i) and torch.equal(s8, full_s)
    # oracle on a sample of the queries against all 8 M keys
    rows = torch.tensor([0, 1, 77, 2047], device=dev)
    rs, ri = cref.topk_cosine(q[rows].cpu().numpy(), kn.cpu().numpy(), k)
    assert np.array_equal(full_i[rows].cpu().numpy(), ri) and np.array_equal(full_s[rows].cpu().numpy(), rs)


# ---------------------------------------------------------------------------------------------------------------------
def _node_model_from(g, dev, noise):
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph

    F_in, D, C = g["X"].shape[1], g["W"].shape[0], g["labels"].shape[1]
    pre = PrePrompt(F_in, D, "prelu", 1, 0.3).to(dev)
    model = RAGraph(pre, None, F_in, C, D, finetune=True, noise_finetune=noise, device=dev)
    with torch.no_grad():
        conv = pre.gcn.convs[0]
        conv.fc.weight.copy_(T(g["W"], dev))
        conv.bias.copy_(T(g["bias"], dev))
        conv.act.weight.copy_(T(g["alpha"], dev))
        model.decoder.fc1.weight.copy_(T(g["fc1_w"], dev))
        model.decoder.fc1.bias.copy_(T(g["fc1_b"], dev))
        model.decoder.fc2.weight.copy_(T(g["fc2_w"], dev))
        model.decoder.fc2.bias.copy_(T(g["fc2_b"], dev))
    model.toy_graph_base.add_resources(T(g["keys"], dev), T(g["values"], dev), T(g["labels"], dev))
    return pre, model


def test_node_retrieve_add_noise_g11a(dev):
    from ragraph_amd.graph import as_csr

    g = gold("g11a_node_noise")
    pre, model = _node_model_from(g, dev, noise=True)
    tgb = model.toy_graph_base
    k, nz = int(g["retrieve_num"]), int(g["noise_retrieve_num"])
    assert tgb.retrieve_num == k and tgb.noise_retrieve_num == nz
    adj = as_csr(T(g["adj"], dev))
    with torch.no_grad():
        h = pre.inference(T(g["X"], dev), adj)
        assert np.allclose(h.cpu().numpy(), g["H"], atol=1e-5)
        hq = T(g["H"], dev)   # the reference's own queries: the fixture's gaps were checked on them
        torch.manual_seed(int(g["seed"]))
        e, l = tgb.retrieve(hq, adj, True)
    assert e.shape == g["rag_embeddings"].shape and l.shape == g["rag_labels"].shape      # [n, 2k + 1, D]
    # the reference draws the noise rows from the default CPU generator: same seed, same rows
    assert np.array_equal(e.cpu().numpy(), g["rag_embeddings"])
    assert np.array_equal(l.cpu().numpy(), g["rag_labels"])
    # deterministic prefix: the first 2k columns are the plain top-2k
    _, idx2k = tgb.topk(hq, 2 * k)
    assert np.array_equal(e[:, :2 * k].cpu().numpy(), T(g["values"], dev)[idx2k].cpu().numpy())
    # and the noisy training-mode forward (RAGraph.py:42-57 with add_noise)
    model.train()
    with torch.no_grad():
        torch.manual_seed(int(g["seed"]))
        logits = model(T(g["X"], dev), adj)
    assert np.allclose(logits.cpu().numpy(), g["train_logits"], atol=2e-5)


def test_graph_retrieve_add_noise_g11b(dev):
    from ragraph_amd.ragraph_utils import ToyGraphBase

    g = gold("g11b_graph_noise")
    C, D = g["labels"].shape[1], g["keys"].shape[1]
    tgb = ToyGraphBase(None, C, D, 1, device=dev, flavour="graph")
    tgb.add_resources(T(g["keys"], dev), T(g["values"], dev), T(g["labels"], dev))
    assert tgb.retrieve_num == int(g["retrieve_num"]) and abs(tgb.noise_std - float(g["noise_std"])) < 1e-9
    torch.manual_seed(int(g["seed"]))
    e, l = tgb.retrieve(T(g["Q"], dev), None, True)
    assert e.shape == g["rag_embeddings"].shape and l.shape == g["rag_labels"].shape      # [1, 2k, D]
    assert np.array_equal(l.cpu().numpy(), g["rag_labels"])
    assert np.array_equal(e.cpu().numpy(), g["rag_embeddings"])   # same CPU-generator noise, one fp32 add per element
    e0, _ = tgb.retrieve(T(g["Q"], dev), None, False)
    assert np.array_equal(e0.cpu().numpy(), T(g["values"], dev)[tgb.topk(T(g["Q"], dev), tgb.retrieve_num)[1]].cpu().numpy())
    d = (e[:, :tgb.retrieve_num] - e0).abs().max().item()
    assert 0 < d < 6 * tgb.noise_std


def test_edge_forward_add_noise_g11c(dev):
    from ragraph_amd.RAGraph_edge import RAGraph as RAGraphEdge

    g = gold("g11c_edge_noise")
    U, I = int(g["num_users"]), int(g["num_items"])

    class DS:
        num_users, num_items = U, I
        edges, edge_norm, edge_times = T(g["edges"], dev), T(g["edge_norm"], dev), T(g["edge_times"], dev)

    class Pre:
        def generate(self):
            return T(g["user_embedding"], dev), T(g["item_embedding"], dev)

    m = RAGraphEdge(DS, Pre(), phase="finetune", use_RAG=True, use_noise=True, retrieve_num=int(g["retrieve_num"]),
                    retrieve_weight=float(g["retrieve_weight"]), batch_size=100, device=dev)
    with torch.no_grad():
        m.gating_weight.copy_(T(g["gating_weight"], dev))
        m.gating_bias.copy_(T(g["gating_bias"], dev))
    assert np.allclose(m.resource_keys.cpu().numpy(), g["resource_keys"], atol=1e-6)
    assert m.noise_retrieve_num == int(g["noise_retrieve_num"])
    m.train()
    with torch.no_grad():
        torch.manual_seed(int(g["seed"]))
        uo, io = m.forward(m.edges, m.edge_norm, m.edge_times)
    ok = g["row_gap"] > 1e-5   # rows whose top-(k+1) indices are well defined (smoothed embeddings tie now and then)
    out = torch.cat([uo, io]).cpu().numpy()
    ref = np.concatenate([g["user_out"], g["item_out"]])
    assert ok.mean() > 0.9
    assert np.allclose(out[ok], ref[ok], atol=1e-5)
    # without the noise column the result differs (the branch really ran) ...
    m.eval()
    with torch.no_grad():
        uo2, io2 = m.forward(m.edges, m.edge_norm, m.edge_times)
    assert not np.allclose(torch.cat([uo2, io2]).cpu().numpy()[ok], ref[ok], atol=1e-5)
    # ... and one [n, 1] draw equals the reference's per-slab draws (three slabs of 100 here)
    torch.manual_seed(int(g["seed"]))
    assert np.array_equal(torch.randint(0, U + I, (U + I, 1)).numpy(), g["noise_idx"])


# ---------------------------------------------------------------------------------------------------------------------
def test_averageemb_class_means(dev):
    """downprompt.averageemb: the class means the reference intends (its own code averages an uninitialised buffer,
    RAGraph_graph/downprompt.py:59-94 -- documented deviation), against a numpy restatement; empty classes give zeros."""
    from ragraph_amd import downprompt as dp

    rng = np.random.default_rng(5)
    emb = rng.standard_normal((57, 256)).astype(np.float32)
    lab = rng.integers(0, 6, 57)
    lab[lab == 4] = 1  # class 4 stays empty
    out = dp.averageemb(T(lab, dev), T(emb, dev), 6).cpu().numpy()
    for c in range(6):
        rows = emb[lab == c]
        if len(rows) == 0:
            assert np.all(out[c] == 0)
            continue
        acc = np.zeros(256, np.float32)
        for r in rows:          # sequential fp32 adds in row order, then one division (segment_reduce's contract)
            acc = acc + r
        assert np.array_equal(out[c], acc / np.float32(len(rows)))
