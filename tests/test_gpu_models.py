"""The host-side mirror of the reference's call surface (ragraph_amd.*), driven on the GPU, against
  (a) the golden vectors the reference itself produced (tests/golden, made by oracle/make_golden.py), and
  (b) the CPU oracle's whole-forward restatements (oracle/pipeline.py) -- bit-exact where no expf is involved.
These tests read like the reference's own usage: build PrePrompt / RAGraph, call model(features, adj).
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


def _load_encoder(pre, g, dev):
    conv = pre.gcn.convs[0]
    with torch.no_grad():
        conv.fc.weight.copy_(T(g["W"], dev))
        conv.bias.copy_(T(g["bias"], dev))
        conv.act.weight.copy_(T(g["alpha"], dev))


def _load_decoder(dec, g, dev):
    with torch.no_grad():
        dec.fc1.weight.copy_(T(g["fc1_w"], dev)); dec.fc1.bias.copy_(T(g["fc1_b"], dev))
        dec.fc2.weight.copy_(T(g["fc2_w"], dev)); dec.fc2.bias.copy_(T(g["fc2_b"], dev))


def test_similarity_functions_g1(dev):
    from ragraph_amd.ragraph_utils import SimilarityFunctions

    g = gold("g1a_cosine_topk")
    S = SimilarityFunctions.calculate_cosine_similarity(T(g["Q"][:8], dev), T(g["K"], dev))
    assert np.allclose(S.cpu().numpy(), g["scores_full_first8"], atol=2e-6)
    s1 = SimilarityFunctions.calculate_cosine_similarity(T(g["Q"][0], dev), T(g["K"], dev))  # 1-D query (graph flavour)
    assert s1.shape == (g["K"].shape[0],) and torch.equal(s1, S[0])
    ts, ti = torch.topk(S, 10)
    assert np.array_equal(ti.cpu().numpy(), g["topk_idx_k10"][:8])


def test_gcn_layer_g4_and_propagation_g5(dev):
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.ragraph_utils import Propagation

    g = gold("g4_gcn_layer")
    pre = PrePrompt(g["X"].shape[1], 256, "prelu", 1, 0.3).to(dev)
    _load_encoder(pre, g, dev)
    adj = T(g["adj"], dev)
    H = pre.inference(T(g["X"], dev), adj)                       # dense adjacency, exactly the reference's call
    assert np.allclose(H.cpu().numpy(), g["H"], atol=1e-5)
    H2 = pre.inference(T(g["X"], dev), CSRGraph.from_dense(adj))  # CSR input: same numbers
    assert torch.equal(H, H2)
    ref = pipeline.gcn_layer(g["X"], cref.dense_to_csr(g["adj"]), g["W"], g["bias"], g["alpha"][0])
    assert np.array_equal(H.cpu().numpy(), ref)                  # bit-exact vs the oracle
    h, c = pre.embed(T(g["X"], dev), adj, False, None, False)
    assert torch.equal(h, H) and c.shape == (1, 256)

    g5 = gold("g5_propagation")
    for k in (0, 1, 2, 3):
        y = Propagation.aggregate_k_hop_features(T(g5["adj"], dev), T(g5["x"], dev), k)
        assert np.allclose(y.cpu().numpy(), g5[f"y_k{k}"], atol=1e-5)
        assert np.array_equal(y.cpu().numpy(), pipeline.propagate(cref.dense_to_csr(g5["adj"]), g5["x"], k))


def test_gcn_layer_bag_of_words_features_take_the_sparse_product(dev, monkeypatch):
    """Cora-shaped features (1.3 % non-zeros over 1433 columns): the layer multiplies X's non-zeros only (CSR SpMM against
    W^T) -- the dense kernel's fmaf chain with the zero terms left out, i.e. the same bits as the dense path and as the
    oracle; dense or narrow feature matrices keep the dense kernel; the CSR form is cached per tensor version."""
    from ragraph_amd import kernels as K
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.layers import gcn as G

    rng = np.random.default_rng(5)
    n, F, D = 700, 1433, 128
    X = (rng.random((n, F)) < 0.013).astype(np.float32) * rng.integers(1, 4, (n, F)).astype(np.float32)
    X[3] = 0.0                                                   # a node without any word
    X[5, :] = (rng.random(F) < 0.5).astype(np.float32)           # and a long row
    adj = (rng.random((n, n)) < 0.01).astype(np.float32)
    adj = np.maximum(adj, adj.T) + np.eye(n, dtype=np.float32)
    adj = adj / adj.sum(1, keepdims=True)
    layer = G.GCN(F, D).to(dev)
    with torch.no_grad():
        layer.bias.copy_(T(rng.standard_normal(D).astype(np.float32), dev))
    g = CSRGraph.from_dense(T(adj, dev))
    Xd = T(X, dev)
    calls = []
    real = K.linear
    monkeypatch.setattr(K, "linear", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    with torch.no_grad():
        assert layer((Xd, g)) is not None and calls == [1]       # a layer never probes its input itself ...
        calls.clear()
        assert G.sparse_features(Xd) is not None                 # ... the encoder's entry point does (PrePrompt)
        H = layer((Xd.unsqueeze(0).squeeze(0), g))               # (a fresh view of the judged tensor, as GcnLayers passes)
        assert calls == []
        ref = pipeline.gcn_layer(X, cref.dense_to_csr(adj), layer.fc.weight.detach().cpu().numpy(),
                                 layer.bias.detach().cpu().numpy(), 0.25)
        assert np.array_equal(H.cpu().numpy(), ref)              # the oracle's dense chain, bit for bit
        monkeypatch.setattr(G, "SPARSE_FEATURES_MAX_DENSITY", 0.0)
        Xd2 = Xd.clone()
        assert G.sparse_features(Xd2) is None
        H2 = layer((Xd2, g))                                     # the same numbers on the dense kernel
        assert calls == [1] and torch.equal(H, H2)
        monkeypatch.setattr(G, "SPARSE_FEATURES_MAX_DENSITY", 0.05)
        Xd.add_(1.0)                                             # a new version of the tensor: judged again -- dense now
        assert G.sparse_features(Xd) is None
        H3 = layer((Xd, g))
        assert calls == [1, 1]
        assert np.array_equal(H3.cpu().numpy(), pipeline.gcn_layer(X + 1.0, cref.dense_to_csr(adj),
                                                                   layer.fc.weight.detach().cpu().numpy(),
                                                                   layer.bias.detach().cpu().numpy(), 0.25))
        layer.fc.weight.mul_(0.5)                                # W^T follows the parameter's version
        Xs = T(X, dev)
        assert G.sparse_features(Xs) is not None
        H4 = layer((Xs, g))
        assert calls == [1, 1]
        assert np.array_equal(H4.cpu().numpy(), pipeline.gcn_layer(X, cref.dense_to_csr(adj),
                                                                   layer.fc.weight.detach().cpu().numpy(),
                                                                   layer.bias.detach().cpu().numpy(), 0.25))


def test_node_forward_g6(dev):
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph

    g = gold("g6_node_forward")
    F_in, C = g["X"].shape[1], g["labels"].shape[1]
    pre = PrePrompt(F_in, 256, "prelu", 1, 0.3).to(dev)
    _load_encoder(pre, g, dev)
    model = RAGraph(pre, None, F_in, C, 256, finetune=True, noise_finetune=False, device=dev)
    _load_decoder(model.decoder, g, dev)
    model.toy_graph_base.add_resources(T(g["keys"], dev), T(g["values"], dev), T(g["labels"], dev))
    model.eval()
    assert model.toy_graph_base.retrieve_num == int(g["k"])
    X, adj = T(g["X"], dev), T(g["adj"], dev)
    with torch.no_grad():
        logits = model(X, adj)
        h = pre.inference(X, adj)
        e, l = model.toy_graph_base.retrieve(h, adj, False)
        dec = model.decoder(T(g["dec_in"], dev))
    assert np.allclose(logits.cpu().numpy(), g["logits"], atol=1e-5)
    assert np.array_equal(e.cpu().numpy(), g["rag_embeddings"]) and np.array_equal(l.cpu().numpy(), g["rag_labels"])
    assert np.allclose(dec.cpu().numpy(), g["dec_out"], atol=1e-5)
    _, idx = model.toy_graph_base.topk(h, int(g["k"]))
    assert np.array_equal(idx.cpu().numpy(), g["topk_idx"])
    # vs the oracle's restatement of RAGraph.forward: same indices, logits to expf rounding
    p = {k: g[k] for k in ("W", "bias", "fc1_w", "fc1_b", "fc2_w", "fc2_b")}
    p["alpha"] = g["alpha"][0]
    ol, oi, oh = pipeline.node_forward(g["X"], cref.dense_to_csr(g["adj"]), p, g["keys"], g["values"], g["labels"],
                                       int(g["k"]), int(g["hops"]), 0.5, 0.5)
    assert np.array_equal(h.cpu().numpy(), oh) and np.array_equal(idx.cpu().numpy(), oi)
    assert np.allclose(logits.cpu().numpy(), ol, atol=1e-6)
    # finetune=False branch returns the mean retrieved label (RAGraph.py:60-63)
    model.finetune = False
    with torch.no_grad():
        lab = model(X, adj)
    assert np.allclose(lab.cpu().numpy(), g["rag_labels"].mean(1), atol=1e-6)


def test_duplicate_keys_g3(dev):
    from ragraph_amd.ragraph_utils import ToyGraphBase

    g = gold("g3_duplicate_keys")
    tgb = ToyGraphBase(None, g["labels"].shape[1], 256, 3, device=dev)
    tgb.add_resources(T(g["keys"], dev), T(g["values"], dev), T(g["labels"], dev))
    sv, ml, idx = tgb.retrieve_reduced(T(g["Q"], dev), int(g["k"]))
    assert np.allclose(sv.cpu().numpy(), g["sum_values"], atol=1e-5)
    assert np.allclose(ml.cpu().numpy(), g["mean_labels"], atol=1e-6)
    _, oi = cref.topk_cosine(g["Q"], cref.normalize_rows(g["keys"]), int(g["k"]))
    assert np.array_equal(idx.cpu().numpy(), oi)


def test_graph_forward_g7(dev):
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraphGraph

    g = gold("g7_graph_forward")
    C = g["labels"].shape[1]
    pre = PrePrompt(1, 256, "prelu", 1, 0.3).to(dev)
    _load_encoder(pre, g, dev)
    model = RAGraphGraph(pre, None, 1, C, 256, device=dev)
    _load_decoder(model.decoder, g, dev)
    model.toy_graph_base.add_resources(T(g["keys"], dev), T(g["values"], dev), T(g["labels"], dev))
    model.eval()
    assert model.toy_graph_base.retrieve_num == int(g["k"])
    X, adj = T(g["X"], dev), T(g["adj"], dev)
    with torch.no_grad():
        logits = model(X, adj)
        h = pre.inference(X, adj)
        e, l = model.toy_graph_base.retrieve(h.mean(dim=0), adj, False)   # 1-D query as the reference passes it
    assert np.allclose(logits.cpu().numpy(), g["logits"], atol=1e-5)
    assert e.shape == g["rag_embeddings"].shape
    assert np.array_equal(l.cpu().numpy(), g["rag_labels"])
    assert np.allclose(e.cpu().numpy(), g["rag_embeddings"], atol=0)


def test_edge_generate_g9(dev):
    from ragraph_amd.RAGraph_edge import RAGraph as RAGraphEdge

    g = gold("g9_edge_generate")
    U, I = int(g["num_users"]), int(g["num_items"])

    class DS:
        num_users, num_items = U, I
        edges, edge_norm, edge_times = T(g["edges"], dev), T(g["edge_norm"], dev), T(g["edge_times"], dev)

    class Pre:
        def generate(self):
            return T(g["user_embedding"], dev), T(g["item_embedding"], dev)

    model = RAGraphEdge(DS, Pre(), phase="finetune", use_RAG=True, retrieve_num=10, device=dev)
    with torch.no_grad():
        model.gating_weight.copy_(T(g["gating_weight"], dev))
        model.gating_bias.copy_(T(g["gating_bias"], dev))
    model.eval()
    # the bank the reference built from the same embeddings: keys = 3x aggregated, values = even layers
    assert np.allclose(model.resource_keys.cpu().numpy(), g["resource_keys"], atol=1e-6)
    assert np.allclose(model.resource_values.cpu().numpy(), g["resource_values"], atol=1e-6)
    tn = model._relative_edge_time_encoding(model.edges, model.edge_times)
    assert np.allclose(tn.cpu().numpy(), g["time_norm"], atol=1e-6)
    with torch.no_grad():   # (outside no_grad the gate is on the autograd tape: the fine-tuning path)
        gated = model.emb_gate(torch.cat([model.user_embedding, model.item_embedding]).detach())
    assert np.allclose(gated.cpu().numpy(), g["gated_emb"], atol=1e-6)
    uo, io = model.generate()
    out = torch.cat([uo, io]).cpu().numpy()
    ref = np.concatenate([g["user_out"], g["item_out"]])
    ok = g["row_gap"] > 1e-5
    assert np.allclose(out[ok], ref[ok], atol=1e-5)
    assert np.allclose(out, ref, atol=5e-3)
    # bit-exact against the oracle's restatement fed the same gated embeddings and bank
    o_out, o_idx, *_ = pipeline.edge_forward(g["edges"], g["edge_norm"], g["edge_times"], gated.cpu().numpy(),
                                             model.resource_keys.cpu().numpy(), model.resource_values.cpu().numpy(),
                                             10, float(g["retrieve_weight"]), 3)
    assert np.allclose(out, o_out, atol=1e-6)

    # query sharding (c5's multi-GPU layout): every rank's slice, stitched the way QueryShard.gather_rows does, gives the
    # unsharded result bit for bit (the collective itself is covered by tests/test_cpu_distributed.py under gloo)
    from ragraph_amd.sharded import shard_bounds

    class Rank:
        def __init__(self, world, rank, parts):
            self.world, self.rank, self.parts = world, rank, parts

        def bounds(self, B):
            return shard_bounds(B, self.world, self.rank)

        def gather_rows(self, local, B):
            self.parts[self.rank] = local.clone()
            lo, hi = self.bounds(B)
            full = local.new_zeros((B,) + tuple(local.shape[1:]))
            full[lo:hi] = local
            return full

    parts, outs = {}, []
    for r in range(3):
        model.query_shard = Rank(3, r, parts)
        outs.append(torch.cat(model.generate()))
    model.query_shard = None
    full = torch.cat([uo, io])
    for r in range(3):
        lo, hi = shard_bounds(U + I, 3, r)
        assert torch.equal(outs[r][lo:hi], full[lo:hi])
    assert sum(p.shape[0] for p in parts.values()) == U + I


def test_downprompt_g10(dev):
    from ragraph_amd import downprompt as dp

    g = gold("g10_downprompt")
    m = dp.downprompt(None, None, None, 256, 2).to(dev)
    with torch.no_grad():
        m.downprompt.weight.copy_(T(g["w"], dev))
    emb = m(T(g["h"], dev), T(g["graph_len"], dev))
    assert np.allclose(emb.cpu().numpy(), g["graph_emb"], atol=1e-4)
    for C in (2, 6):
        lp = dp.predict(emb.shape[0], C, emb, T(g[f"proto_c{C}"], dev))
        assert np.allclose(lp.cpu().numpy(), g[f"logp_c{C}"], atol=1e-5)


def test_downprompt_node_g16(dev):
    """Node flavour (RAGraph_node/downprompt.py:6-48,59-78,80-130) against the reference's own outputs and the oracle."""
    from ragraph_amd import downprompt_node as dpn
    from ragraph_amd import kernels as K

    g = gold("g16_downprompt_node")
    pr = T(g["prompt"], dev)
    m = dpn.downprompt(pr[0:1], pr[1:2], pr[2:3], 256, 3, T(g["feature"], dev).unsqueeze(0), T(g["labels"], dev)).to(dev)
    with torch.no_grad():
        m.downprompt.weight.copy_(T(g["w"], dev))
    h = T(g["h"], dev)
    o_ave = pipeline.downprompt_node_averageemb(g["labels"], g["feature"])
    assert np.array_equal(m.ave.cpu().numpy(), o_ave)                           # bit-exact vs the oracle
    assert np.allclose(m.ave.cpu().numpy(), g["ave_init"], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        raw = m.downprompt(h)
        probs = m(h, train=0)
    o_probs, o_raw = pipeline.downprompt_node_forward(g["h"], g["w"], o_ave)
    assert np.allclose(raw.cpu().numpy(), o_raw, atol=1e-6) and np.allclose(raw.cpu().numpy(), g["elu_wh"], atol=1e-6)
    assert np.array_equal(np.maximum(raw.cpu().numpy(), 0), np.maximum(o_raw, 0))   # positive side: no expm1, same bits
    assert np.allclose(probs.cpu().numpy(), o_probs, atol=1e-6)
    assert np.allclose(probs.cpu().numpy(), g["probs"], atol=1e-6)
    m.ave = T(g["ave_injected"], dev)
    with torch.no_grad():
        assert np.allclose(m(h, train=0).cpu().numpy(), g["probs_injected"], atol=1e-6)
        assert np.allclose(m(h, train=1).cpu().numpy(), g["probs_train"], atol=1e-6)
    assert np.allclose(m.ave.cpu().numpy(), g["ave_train"], rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        assert np.allclose(m.nodelabelprompt(m.prompt).cpu().numpy(), g["weighted_prompt"], atol=1e-6)
        assert np.allclose(m.dffprompt(h, T(g["feature"], dev)).cpu().numpy(), g["weighted_feature"], atol=1e-6)
    # a class that does not fit the reference's [3, n/2, D] buffer: IndexError there, IndexError here
    with pytest.raises(IndexError):
        dpn.averageemb(torch.zeros(10, dtype=torch.int64, device=dev), h[:10])


@pytest.mark.parametrize("D", [256, 30, 7])
def test_averageemb_any_width(dev, D):
    """Graph flavour's class means for embedding widths that are not multiples of 4 (round 2 returned zeros)."""
    from ragraph_amd import downprompt as dp

    rng = np.random.default_rng(D)
    x = rng.standard_normal((57, D)).astype(np.float32)
    lab = rng.integers(0, 4, 57)
    lab[lab == 2] = 3                                                          # class 2 is empty
    out = dp.averageemb(T(lab, dev), T(x, dev), 5).cpu().numpy()
    for c in range(5):
        rows = x[lab == c]
        if len(rows) == 0:
            assert not out[c].any()
            continue
        acc = np.zeros(D, dtype=np.float32)
        for r in rows:
            acc = acc + r
        assert np.array_equal(out[c], acc / np.float32(len(rows)))


def test_bank_build_and_finetune_step(dev):
    """The reference's driver loop in miniature (finetune-rag.py:57-84): build the bank from a resource dataset, one
    Adam step on the decoder, loss decreases; retrieving a stored key returns that key first."""
    from ragraph_amd.data import DataLoader, synthetic_tu_dataset
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph
    from ragraph_amd.ragraph_utils import process_tu_dataset, seed_everything

    seed_everything(0)
    ds = synthetic_tu_dataset(num_graphs=12, num_node_attributes=18, num_node_labels=3, seed=9)
    pre = PrePrompt(18, 256, "prelu", 1, 0.3).to(dev)
    model = RAGraph(pre, ds[:6], 18, 3, 256, finetune=True, device=dev)
    tgb = model.toy_graph_base
    assert tgb.resource_keys.shape == (6 * 4 * 10, 256)       # 6 graphs x (1 + 3 augmentations) x 10 samples
    assert tgb.resource_labels.shape[1] == 3 and torch.isfinite(tgb.resource_values).all()
    # rows 0..9 come from the un-augmented graph; the augmented copies multiply their features by a Bernoulli mask of
    # probability sample_prob * 0.01 (Augmentation.py:17-18), i.e. almost surely zero -> zero keys, as in the reference
    s, _ = tgb.topk(tgb.resource_keys[:10], 1)
    assert torch.allclose(s, torch.ones_like(s), atol=1e-5)   # a stored key's best match has cosine 1
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    model.train()
    losses = []
    batch = next(iter(DataLoader(ds[6:], batch_size=6)))
    feats, adj, labels = process_tu_dataset(batch, 18, device=dev)
    for _ in range(5):
        opt.zero_grad()
        out = model(feats, adj)
        loss = torch.nn.functional.cross_entropy(out, labels.argmax(1))
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(p.grad is not None for p in model.decoder.parameters())
    assert losses[-1] < losses[0]


def test_decoder_gradients_match_torch(dev):
    from ragraph_amd import autograd as A
    from ragraph_amd import kernels as K

    torch.manual_seed(0)
    x = torch.randn(37, 256, device=dev, requires_grad=True)
    w1 = (torch.randn(256, 256, device=dev) / 16).requires_grad_()
    b1 = torch.randn(256, device=dev, requires_grad=True)
    w2 = (torch.randn(3, 256, device=dev) / 16).requires_grad_()
    b2 = torch.randn(3, device=dev, requires_grad=True)
    rl = torch.nn.functional.one_hot(torch.randint(0, 3, (37,), device=dev), 3).float()
    tgt = torch.randint(0, 3, (37,), device=dev)

    def run(lin, smx):
        h = lin(x, w1, b1, True)
        return torch.nn.functional.nll_loss(torch.log(smx(lin(h, w2, b2, False), rl)), tgt)

    ours = run(lambda a, w, b, act: A.linear(a, w, b, act=K.ACT_LEAKY if act else K.ACT_NONE, alpha=0.01),
               lambda lg, r: A.softmax_mix(lg, r, 0.5))
    g_ours = torch.autograd.grad(ours, [x, w1, b1, w2, b2])
    ref = run(lambda a, w, b, act: torch.nn.functional.leaky_relu(torch.nn.functional.linear(a, w, b), 0.01) if act
              else torch.nn.functional.linear(a, w, b), lambda lg, r: torch.softmax(lg, 1) * 0.5 + r * 0.5)
    g_ref = torch.autograd.grad(ref, [x, w1, b1, w2, b2])
    assert torch.allclose(ours, ref, atol=1e-5)
    for a, b in zip(g_ours, g_ref):
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-4)


def test_graph_flavour_batched_equals_per_graph(dev):
    """Config 3 (PROTEINS-style batches): one batched pass over 16 graphs == 16 single-graph forwards, bit for bit."""
    from ragraph_amd.data import DataLoader, synthetic_tu_dataset
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraphGraph

    torch.manual_seed(0)
    ds = synthetic_tu_dataset(num_graphs=16, num_node_attributes=4, num_node_labels=3, num_classes=2, seed=5)
    pre = PrePrompt(4, 256, "prelu", 1, 0.3).to(dev)
    model = RAGraphGraph(pre, None, 4, 2, 256, device=dev).eval()
    N = 3000
    model.toy_graph_base.add_resources(torch.nn.functional.normalize(torch.randn(N, 256, device=dev), dim=-1),
                                       torch.randn(N, 256, device=dev),
                                       torch.nn.functional.one_hot(torch.randint(0, 2, (N,), device=dev), 2).float())
    batch = next(iter(DataLoader(ds, batch_size=16)))
    X = batch.x[:, :4].to(dev)
    adj = CSRGraph.from_edge_index_sym_normalized(batch.edge_index.to(dev), X.shape[0])
    with torch.no_grad():
        out_b = model.forward_batch(X, adj, batch.ptr)
        singles = []
        for gi in range(16):
            lo, hi = int(batch.ptr[gi]), int(batch.ptr[gi + 1])
            g = batch[gi]
            a = CSRGraph.from_edge_index_sym_normalized(g.edge_index.to(dev), hi - lo)
            singles.append(model(X[lo:hi].contiguous(), a))
    assert torch.equal(out_b, torch.cat(singles))


def test_forward_is_hip_graph_capturable(dev):
    """The C ABI allocates nothing and never synchronises, so a whole small forward (a dozen launches, launch-bound at
    the reference's scale) can be captured once into a HIP graph and replayed.  The bank is large enough for the
    bf16-filtered retrieval, which repairs overflowed rows on the device and reads nothing back: it is captured too."""
    from ragraph_amd.data import DataLoader, synthetic_tu_dataset
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph
    from ragraph_amd.ragraph_utils import process_tu_dataset

    torch.manual_seed(0)
    ds = synthetic_tu_dataset(num_graphs=16, num_node_attributes=18, num_node_labels=3, seed=2)
    pre = PrePrompt(18, 256, "prelu", 1, 0.3).to(dev)
    model = RAGraph(pre, None, 18, 3, 256, device=dev).eval()
    N = 70000
    model.toy_graph_base.add_resources(torch.nn.functional.normalize(torch.randn(N, 256, device=dev), dim=-1),
                                       torch.randn(N, 256, device=dev),
                                       torch.nn.functional.one_hot(torch.randint(0, 3, (N,), device=dev), 3).float())
    feats, adj, _ = process_tu_dataset(next(iter(DataLoader(ds, batch_size=16))), 18, device=dev)
    _ = adj.row_normalized_values()
    with torch.no_grad():
        eager = model(feats, adj)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                model(feats, adj)  # warm-up on the capture stream (workspace, LDS attributes)
        torch.cuda.current_stream().wait_stream(s)
        from ragraph_amd import kernels as K
        calls = []
        orig = K.topk_cosine_filtered
        K.topk_cosine_filtered = lambda *a, **kw: (calls.append(1), orig(*a, **kw))[1]
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                captured = model(feats, adj)
        finally:
            K.topk_cosine_filtered = orig
        assert calls, "the captured forward must contain the bf16-filtered retrieval"
        feats.mul_(1.0)  # same static input buffers
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(captured, eager)


def test_captured_forward_helper_replays_new_inputs(dev):
    """ragraph_amd.capture.CapturedForward: capture a batched graph-classification forward once, replay it on OTHER
    features of the same shape -- the replayed output equals the eager forward on those features, bit for bit."""
    from ragraph_amd.capture import CapturedForward
    from ragraph_amd.data import DataLoader, synthetic_tu_dataset
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraphGraph
    from ragraph_amd.ragraph_utils import process_tu_dataset

    torch.manual_seed(1)
    ds = synthetic_tu_dataset(num_graphs=16, num_node_attributes=18, num_node_labels=3, seed=4)
    model = RAGraphGraph(PrePrompt(18, 128, "prelu", 1, 0.3).to(dev), None, 18, 2, 128, device=dev).eval()
    model.toy_graph_base.add_resources(torch.nn.functional.normalize(torch.randn(1113, 128, device=dev), dim=-1),
                                       torch.randn(1113, 128, device=dev),
                                       torch.nn.functional.one_hot(torch.randint(0, 2, (1113,), device=dev), 2).float())
    data = next(iter(DataLoader(ds, batch_size=16)))
    feats, adj, _ = process_tu_dataset(data, 18, device=dev)
    ptr = data.ptr.to(dev)
    _ = adj.row_normalized_values()
    fwd = CapturedForward(lambda x: model.forward_batch(x, adj, ptr), feats)
    other = torch.rand_like(feats)
    with torch.no_grad():
        want = model.forward_batch(other, adj, ptr)
        want0 = model.forward_batch(feats, adj, ptr)
    got = fwd(other).clone()
    assert torch.equal(got, want)
    assert torch.equal(fwd(feats), want0)
    with pytest.raises(ValueError):
        fwd(other[:-1])


def test_key_sharded_forward_with_query_sharded_tail_one_rank_rccl(dev):
    """RAGraph._forward_key_shard under a 1-rank RCCL group (the collectives run: all_reduce / all_gather of the theta
    exchange, all_to_all of the lists, all_gather of the outputs): bit-identical to the plain forward."""
    import socket

    import torch.distributed as dist

    from ragraph_amd.data import synthetic_big_graph
    from ragraph_amd.graph import CSRGraph
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph import RAGraph
    from ragraph_amd.sharded import ShardedToyGraphBase

    torch.manual_seed(3)
    n, F, D, C, N, k = 3000, 32, 128, 3, 70000, 10
    model = RAGraph(PrePrompt(F, D, "prelu", 1, 0.3).to(dev), None, F, C, D, device=dev).eval()
    model.toy_graph_base.retrieve_num = k
    keys = torch.nn.functional.normalize(torch.randn(N, D, device=dev), dim=-1)
    vals = torch.randn(N, D, device=dev)
    labs = torch.nn.functional.one_hot(torch.randint(0, C, (N,), device=dev), C).float()
    model.toy_graph_base.add_resources(keys, vals, labs)
    adj = CSRGraph.from_edge_index_sym_normalized(synthetic_big_graph(n, 6, seed=2, device=dev), n)
    X = torch.randn(n, F, device=dev)
    with torch.no_grad():
        want = model(X, adj)
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        plain = model.toy_graph_base
        model.toy_graph_base = ShardedToyGraphBase(keys, vals, labs, 0, k, force_collectives=True, values_replicated=True)
        calls = []
        orig = model._forward_key_shard
        model._forward_key_shard = lambda *a: (calls.append(1), orig(*a))[1]
        with torch.no_grad():
            got = model(X, adj)
        assert calls, "the sharded bank must take the key-shard forward"
        model.toy_graph_base = plain
    finally:
        dist.destroy_process_group()
    assert torch.equal(got, want)


def test_bank_save_load_roundtrip(dev, tmp_path):
    from ragraph_amd.ragraph_utils import ToyGraphBase

    tgb = ToyGraphBase(None, 3, 256, 3, device=dev)
    k, v = torch.randn(500, 256, device=dev), torch.randn(500, 256, device=dev)
    l = torch.nn.functional.one_hot(torch.randint(0, 3, (500,), device=dev), 3).float()
    tgb.add_resources(torch.nn.functional.normalize(k, dim=-1), v, l)
    q = torch.randn(9, 256, device=dev)
    a = tgb.retrieve_reduced(q)
    tgb.save(str(tmp_path / "bank.pt"))
    t2 = ToyGraphBase(None, 3, 256, 3, device=dev)
    t2.load(str(tmp_path / "bank.pt"))
    b = t2.retrieve_reduced(q)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    t2.load(str(tmp_path / "bank.pt"), append=True)   # appended duplicates: ties resolve to the lower index
    assert t2.resource_keys.shape[0] == 1000 and torch.equal(t2.retrieve_reduced(q)[2][:, 0], a[2][:, 0])
