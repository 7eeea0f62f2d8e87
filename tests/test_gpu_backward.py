"""SURVEY.md section 8(f) row 4 / section 7.3 hard part 4: fine-tuning backward on the HIP kernels, each gradient against
torch autograd of a plain fp32 eager restatement of the same op (tolerance 1e-4 relative: sums are re-associated), the
graph_fewshot flavour against golden g15 from the reference, and the edge flavour's cal_loss."""
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


def close(a, b, tol=1e-4):
    return float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def _random_graph(dev, n, density, seed):
    g = torch.Generator().manual_seed(seed)
    a = (torch.rand(n, n, generator=g) < density).float() * torch.rand(n, n, generator=g)
    a = a + torch.eye(n) * 0.5
    return a.to(dev)


def test_spmm_autograd_matches_torch(dev):
    from ragraph_amd import autograd as A
    from ragraph_amd import kernels as K
    from ragraph_amd.graph import CSRGraph

    torch.manual_seed(0)
    n, D = 157, 64
    a = _random_graph(dev, n, 0.05, 1)              # NOT symmetric: the backward needs the transposed CSR
    g = CSRGraph.from_dense(a)
    w = torch.randn(n, D, device=dev)
    for act, slope in ((K.ACT_PRELU, 0.2), (K.ACT_RELU, 0.0), (K.ACT_NONE, 0.0)):
        x = torch.randn(n, D, device=dev, requires_grad=True)
        bias = torch.randn(D, device=dev, requires_grad=True)
        alpha = torch.full((1,), slope, device=dev, requires_grad=True) if act == K.ACT_PRELU else None
        y = A.spmm_csr(g, x, bias, act, alpha, slope)
        (y * w).sum().backward()
        x2, b2 = x.detach().clone().requires_grad_(True), bias.detach().clone().requires_grad_(True)
        z = a @ x2 + b2
        if act == K.ACT_PRELU:
            a2 = alpha.detach().clone().requires_grad_(True)
            ref = torch.nn.functional.prelu(z, a2)
        else:
            ref = torch.relu(z) if act == K.ACT_RELU else z
        (ref * w).sum().backward()
        assert close(y.detach(), ref.detach(), 1e-5)
        assert close(x.grad, x2.grad) and close(bias.grad, b2.grad)
        if act == K.ACT_PRELU:
            assert close(alpha.grad, a2.grad)


@pytest.mark.parametrize("slope", [0.25, 0.0, -0.3])
def test_gcn_decode_layer_trains_like_torch(dev, slope):
    """The few-shot flavours' trainable layer (RAGraph_node_fewshot/RAGraph.py:69): fc weight, bias and PReLU slope -- also
    for a slope that training has driven to zero or below (nn.PReLU is unconstrained): the fused epilogue's output no
    longer determines the pre-activation there, and the layer keeps it instead."""
    from ragraph_amd.gcnlayers import GcnLayers
    from ragraph_amd.graph import CSRGraph

    torch.manual_seed(1)
    n, D = 90, 256
    a = _random_graph(dev, n, 0.08, 2)
    a = (a + a.t()) / 2
    g = CSRGraph.from_dense(a)
    net = GcnLayers(18, D, 2, 0.3).to(dev)
    with torch.no_grad():
        net.convs[1].bias.normal_(0, 0.1)
        net.convs[1].act.weight.fill_(slope)
    h = torch.randn(n, D, device=dev)
    w = torch.randn(n, D, device=dev)
    out = net.decode(h, g)
    (out * w).sum().backward()
    c = net.convs[1]
    W2, b2, a2 = (t.detach().clone().requires_grad_(True) for t in (c.fc.weight, c.bias, c.act.weight))
    ref = torch.nn.functional.prelu(a @ (h @ W2.t()) + b2, a2)
    (ref * w).sum().backward()
    assert close(out.detach(), ref.detach(), 1e-4)
    assert close(c.fc.weight.grad, W2.grad) and close(c.bias.grad, b2.grad) and close(c.act.weight.grad, a2.grad)
    assert all(p.grad is None for p in net.convs[0].parameters())      # encode is not on the tape


def test_gate_gather_softmax_grads(dev):
    from ragraph_amd import autograd as A

    torch.manual_seed(2)
    x = torch.randn(300, 64, device=dev, requires_grad=True)
    z = torch.randn(300, 64, device=dev, requires_grad=True)
    w = torch.randn(300, 64, device=dev)
    (A.sigmoid_gate(x, z) * w).sum().backward()
    x2, z2 = x.detach().clone().requires_grad_(True), z.detach().clone().requires_grad_(True)
    ((x2 * torch.sigmoid(z2)) * w).sum().backward()
    assert close(x.grad, x2.grad) and close(z.grad, z2.grad)
    v = torch.randn(50, 64, device=dev, requires_grad=True)
    idx = torch.tensor([3, 3, 7, 49, 0, 3, 7], device=dev)
    wv = torch.randn(7, 64, device=dev)
    (A.gather_rows(v, idx) * wv).sum().backward()
    v2 = v.detach().clone().requires_grad_(True)
    (v2[idx] * wv).sum().backward()
    assert close(v.grad, v2.grad)
    lg = torch.randn(40, 7, device=dev, requires_grad=True)
    rl = torch.rand(40, 7, device=dev)
    ws = torch.randn(40, 7, device=dev)
    (A.softmax_mix(lg, rl, 0.3) * ws).sum().backward()
    l2 = lg.detach().clone().requires_grad_(True)
    ((torch.softmax(l2, 1) * 0.7 + rl * 0.3) * ws).sum().backward()
    assert close(lg.grad, l2.grad)


def test_graph_fewshot_forward_g15_and_training_step(dev):
    from ragraph_amd.graph import as_csr
    from ragraph_amd.preprompt import PrePrompt
    from ragraph_amd.RAGraph_fewshot import RAGraphGraphFewShot

    g = gold("g15_graph_fewshot")
    F_in, D, C = g["X"].shape[1], 256, g["labels"].shape[1]
    pre = PrePrompt(F_in, D, "prelu", 2, 0.3).to(dev)
    with torch.no_grad():
        for i, conv in enumerate(pre.gcn.convs):
            conv.fc.weight.copy_(T(g[f"W{i}"], dev))
            conv.bias.copy_(T(g[f"b{i}"], dev))
            conv.act.weight.copy_(T(g[f"a{i}"], dev))
    model = RAGraphGraphFewShot(pre, None, F_in, C, D, device=dev, dataset_name="PROTEINS").eval()
    model.toy_graph_base.add_resources(T(g["keys"], dev), T(g["values"], dev), T(g["labels"], dev))
    adj = as_csr(T(g["adj"], dev))
    mfl = T(g["mean_fewshot_logits"], dev)
    with torch.no_grad():
        out = model(T(g["X"], dev), adj, mfl)
    assert out.shape == g["logits"].shape and np.allclose(out.cpu().numpy(), g["logits"], atol=2e-5)
    p = {k: g[k] for k in ("W0", "b0", "W1", "b1")}
    p["a0"], p["a1"] = float(g["a0"][0]), float(g["a1"][0])
    o, oidx, oh = pipeline.graph_fewshot_forward(g["X"], cref.dense_to_csr(g["adj"]), p, g["keys"], g["values"], g["labels"],
                                                 g["mean_fewshot_logits"], int(g["k"]), 0.5, 0.5)
    assert np.array_equal(out.cpu().numpy(), o)                         # bit for bit vs the oracle composition
    # the bank builder reproduces the rows the reference built from the two resource graphs
    from ragraph_amd.data import Data, GraphDataset
    graphs = []
    for tag, lab in (("res0", 0), ("res1", 1)):
        a = g[tag + "_adj"]
        ei = np.stack(np.nonzero((a != 0) & ~np.eye(a.shape[0], dtype=bool)))
        graphs.append(Data(torch.from_numpy(g[tag + "_x"]), torch.from_numpy(ei).long(), torch.tensor([lab])))
    m2 = RAGraphGraphFewShot(pre, GraphDataset(graphs, F_in, C, "PROTEINS"), F_in, C, D, device=dev)
    tgb = m2.toy_graph_base
    assert np.allclose(tgb.resource_keys.cpu().numpy(), g["built_keys"], atol=1e-5)
    assert np.allclose(tgb.resource_values.cpu().numpy(), g["built_values"], atol=1e-5)
    assert np.array_equal(tgb.resource_labels.cpu().numpy(), g["built_labels"].astype(np.float32))
    # one fine-tuning step: gradients reach the decode layer only, the loss goes down
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    target = torch.randn(1, D, device=dev)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        loss = ((model(T(g["X"], dev), adj, mfl) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(q.grad is not None for q in pre.gcn.convs[1].parameters())
    assert all(q.grad is None for q in pre.gcn.convs[0].parameters())
    assert losses[-1] < losses[0]


def test_edge_cal_loss_gradients_match_torch(dev):
    """RAGraph_edge fine-tuning step (modules/RAGraph.py:335-355): the gradients of embeddings, gate and LoRA factors
    through gate -> 3 propagation layers -> batch gathers -> BPR + L2, against torch autograd of an eager restatement."""
    from ragraph_amd.data import synthetic_bipartite
    from ragraph_amd.RAGraph_edge import RAGraph as RAGraphEdge

    U, I, D = 300, 200, 64
    edges, norm, times = synthetic_bipartite(U, I, edges_per_user=6, seed=12, device=dev)

    class DS:
        num_users, num_items = U, I
    DS.edges, DS.edge_norm, DS.edge_times = edges, norm, times

    class Pre:
        def generate(self):
            g = torch.Generator(device=dev).manual_seed(5)
            return 0.1 * torch.randn(U, D, device=dev, generator=g), 0.1 * torch.randn(I, D, device=dev, generator=g)

    torch.manual_seed(3)
    m = RAGraphEdge(DS, Pre(), phase="finetune", use_RAG=True, use_LoRA=True, LoRA_rank=8, retrieve_num=5, device=dev).train()
    m.edge_dropout = 0.0                                              # all edges: the restatement below needs no mask
    batch = (torch.randint(0, U, (64,)), torch.randint(0, I, (64,)), torch.randint(0, I, (64,)))
    loss, parts = m.cal_loss(batch)
    loss.backward()
    names = ["user_embedding", "item_embedding", "gating_weight", "gating_bias", "user_embedding_A", "user_embedding_B",
             "item_embedding_A", "item_embedding_B"]
    got = {k: getattr(m, k).grad.clone() for k in names}
    assert all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for v in got.values())
    # eager restatement with torch ops (time weights and the retrieved term taken as constants from the HIP forward)
    P = {k: getattr(m, k).detach().clone().requires_grad_(True) for k in names}
    with torch.no_grad():
        tn = m._relative_edge_time_encoding(edges, times)
        en = norm * 0.5 + tn * 0.5
        m.eval()
        uo, io = m.forward(edges, norm, times)
        m.use_RAG = False
        up, ip = m.forward(edges, norm, times)
        m.use_RAG = True
        rag = (torch.cat([uo, io]) - 0.7 * torch.cat([up, ip])) / 0.3
    ue = P["user_embedding"] + P["user_embedding_A"] @ P["user_embedding_B"]
    ie = P["item_embedding"] + P["item_embedding_A"] @ P["item_embedding_B"]
    x = torch.cat([ue, ie])
    x = x * torch.sigmoid(x @ P["gating_weight"] + P["gating_bias"])
    res = [x]
    for _ in range(3):
        res.append(torch.zeros_like(x).index_add_(0, edges[:, 1], res[-1][edges[:, 0]] * en[:, None]))
    tot = 0.7 * sum(res) + 0.3 * rag
    ueo, ieo = tot[:U], tot[U:]
    us, ps, ns = (t.to(dev) for t in batch)
    pos, neg = (ueo[us] * ieo[ps]).sum(1), (ueo[us] * ieo[ns]).sum(1)
    rec = (-torch.log(1e-10 + torch.sigmoid(pos - neg))).mean()
    reg = 0.5 * (ue[us].norm(2) ** 2 + ie[ps].norm(2) ** 2 + ie[ns].norm(2) ** 2) / 64.0
    ref = rec + 1e-4 * reg
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    for k in names:
        assert close(got[k], P[k].grad, 2e-4), k



@pytest.mark.parametrize("rng", ["host", "device"])
def test_edge_cal_loss_with_edge_dropout_rebuilds_its_graph_natively(dev, rng):
    """The fine-tuning step WITH edge dropout (modules/RAGraph.py:337-343, utils.py:40-53): every step draws a mask, keeps the
    set edges (ragraph_mask_positions_i64) and rebuilds the destination-sorted CSR (ragraph_coo_to_csr_i64) -- against a
    restatement of the same step on the SAME mask with torch ops: the loss to 1e-5, finite gradients on every parameter.  Mask on
    the host generator (the reference's draw) and on the device."""
    from ragraph_amd import kernels as K
    from ragraph_amd.data import synthetic_bipartite
    from ragraph_amd.RAGraph_edge import RAGraph as RAGraphEdge

    U, I, D = 300, 200, 64
    edges, norm, times = synthetic_bipartite(U, I, edges_per_user=6, seed=12, device=dev)

    class DS:
        num_users, num_items = U, I
    DS.edges, DS.edge_norm, DS.edge_times = edges, norm, times

    class Pre:
        def generate(self):
            g = torch.Generator(device=dev).manual_seed(5)
            return 0.1 * torch.randn(U, D, device=dev, generator=g), 0.1 * torch.randn(I, D, device=dev, generator=g)

    torch.manual_seed(3)
    m = RAGraphEdge(DS, Pre(), phase="finetune", use_RAG=False, use_LoRA=False, retrieve_num=5, device=dev).train()
    m.edge_dropout, m.dropout_rng = 0.5, rng
    masks = []
    real = m.draw_edge_mask
    m.draw_edge_mask = lambda: (masks.append(real()), masks[-1])[1]
    batch = (torch.randint(0, U, (64,)), torch.randint(0, I, (64,)), torch.randint(0, I, (64,)))
    loss, _ = m.cal_loss(batch)
    loss.backward()
    mask = masks[0]
    assert mask.is_cuda and mask.dtype == torch.bool and 0.35 < float(mask.float().mean()) < 0.65
    assert torch.equal(K.mask_positions(mask), torch.nonzero(mask).reshape(-1))
    for name in ("user_embedding", "item_embedding", "gating_weight", "gating_bias"):
        gr = getattr(m, name).grad
        assert gr is not None and torch.isfinite(gr).all() and float(gr.abs().max()) > 0, name
    # the same step on the same kept edges, torch ops
    e, w, t = edges[mask], norm[mask], times[mask]
    with torch.no_grad():
        tn = m._relative_edge_time_encoding(e, t)
        en = w * 0.5 + tn * 0.5
        x = torch.cat([m.user_embedding, m.item_embedding])
        x = x * torch.sigmoid(x @ m.gating_weight + m.gating_bias)
        res = [x]
        for _ in range(3):
            res.append(torch.zeros_like(x).index_add_(0, e[:, 1], res[-1][e[:, 0]] * en[:, None]))
        tot = sum(res)
        ueo, ieo = tot[:U], tot[U:]
        us, ps, ns = (b.to(dev) for b in batch)
        pos, neg = (ueo[us] * ieo[ps]).sum(1), (ueo[us] * ieo[ns]).sum(1)
        rec = (-torch.log(1e-10 + torch.sigmoid(pos - neg))).mean()
        reg = 0.5 * (m.user_embedding[us].norm(2) ** 2 + m.item_embedding[ps].norm(2) ** 2 + m.item_embedding[ns].norm(2) ** 2) / 64.0
    assert abs(float(loss) - float(rec + 1e-4 * reg)) < 1e-5


@pytest.mark.parametrize("flavour", ["node", "graph"])
def test_downprompt_weight_gradients_match_torch(dev, flavour):
    """The prompt weight is the trainable parameter of downstreamprompt (RAGraph_node/downprompt.py:118-130 with ELU,
    RAGraph_graph/downprompt.py:154-168 without): its gradient through the cosine-to-prototype head, against torch
    autograd of the reference's formula."""
    import torch.nn.functional as F
    from ragraph_amd import autograd as AG
    from ragraph_amd import downprompt as dpg
    from ragraph_amd import downprompt_node as dpn

    torch.manual_seed(3)
    n, D = 83, 256 if flavour == "node" else 30
    h = torch.randn(n, D, device=dev)
    ave = torch.randn(3, D, device=dev)
    tgt = torch.randint(0, 3, (n,), device=dev)
    mod = (dpn.downstreamprompt(D) if flavour == "node" else dpg.downstreamprompt(D)).to(dev)
    hq = h.clone().requires_grad_(True)
    raw = mod(hq)
    assert raw.grad_fn is not None
    mode = 1 if flavour == "node" else 2
    out = AG.proto_cosine(raw, ave, mode)
    loss = F.nll_loss(torch.log(out) if mode == 1 else out, tgt)
    loss.backward()
    w_ref = mod.weight.detach().clone().requires_grad_(True)
    h_ref = h.clone().requires_grad_(True)
    z = w_ref * h_ref
    raw_ref = F.elu(z) if flavour == "node" else z
    cos = F.cosine_similarity(raw_ref[:, None, :], ave[None, :, :], dim=-1, eps=1e-8)
    out_ref = F.softmax(cos, 1) if mode == 1 else F.log_softmax(cos, 1)
    loss_ref = F.nll_loss(torch.log(out_ref) if mode == 1 else out_ref, tgt)
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    assert torch.allclose(mod.weight.grad, w_ref.grad, atol=1e-5), (mod.weight.grad - w_ref.grad).abs().max()
    assert torch.allclose(hq.grad, h_ref.grad, atol=1e-5)


def test_node_downprompt_training_step_keeps_prototypes_in_the_graph(dev):
    """RAGraph_node/downprompt.py:24-46 with train = 1: the prototypes are averageemb of the step's OWN prompted embeddings and
    stay in the autograd graph, so the prompt weight's gradient has two paths (samples and prototypes).  Against torch autograd
    of the reference's formula with a zero-filled buffer (class sums / floor(n / 2))."""
    import torch.nn.functional as F
    from ragraph_amd import downprompt_node as dpn

    torch.manual_seed(11)
    n, D = 90, 256
    h = torch.randn(n, D, device=dev)
    labels = torch.randint(0, 3, (n,), device=dev)
    tgt = labels.clone()
    prompts = [torch.randn(1, D, device=dev) for _ in range(3)]
    mod = dpn.downprompt(*prompts, D, 3, h.clone(), labels).to(dev)
    out = mod(h, train=1)
    assert mod.ave.grad_fn is not None
    loss = F.nll_loss(torch.log(out), tgt)
    loss.backward()
    w_ref = mod.downprompt.weight.detach().clone().requires_grad_(True)
    raw = F.elu(w_ref * h)
    half = n // 2
    ave = torch.stack([raw[labels == c].sum(0) / half for c in range(3)])
    cos = F.cosine_similarity(raw[:, None, :], ave[None, :, :], dim=-1, eps=1e-8)
    loss_ref = F.nll_loss(torch.log(F.softmax(cos, 1)), tgt)
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    g, gr = mod.downprompt.weight.grad, w_ref.grad
    assert torch.allclose(g, gr, atol=2e-6 + 1e-4 * float(gr.abs().max())), (g - gr).abs().max()
    # the prototype path matters: with detached prototypes the gradient is a different vector
    w2 = mod.downprompt.weight.detach().clone().requires_grad_(True)
    raw2 = F.elu(w2 * h)
    cos2 = F.cosine_similarity(raw2[:, None, :], ave.detach()[None, :, :], dim=-1, eps=1e-8)
    F.nll_loss(torch.log(F.softmax(cos2, 1)), tgt).backward()
    assert (w2.grad - gr).abs().max() > 10 * (g - gr).abs().max()
    # evaluation afterwards: no graph of the past step behind the output
    ev = mod(h, train=0)
    ev.sum().backward()        # (would raise "backward through the graph a second time" if it reached the past step's prototypes)


@pytest.mark.parametrize("G,C,D,mode", [(1000, 3, 256, 1), (257, 5, 64, 2), (64, 1, 30, 0), (5000, 16, 128, 1)])
def test_proto_cosine_gradient_for_the_prototypes(dev, G, C, D, mode):
    """ragraph_proto_cosine_grad_proto_f32 against torch autograd; identical from run to run (fixed summation order)."""
    import torch.nn.functional as F
    from ragraph_amd import kernels as K

    torch.manual_seed(G + C)
    emb = torch.randn(G, D, device=dev)
    proto = torch.randn(C, D, device=dev)
    go = torch.randn(G, C, device=dev)
    out = K.proto_cosine(emb, proto, mode)
    got = K.proto_cosine_grad_proto(emb, proto, mode, out, go)
    assert torch.equal(got, K.proto_cosine_grad_proto(emb, proto, mode, out, go))
    pr = proto.double().clone().requires_grad_(True)
    cos = F.cosine_similarity(emb.double()[:, None, :], pr[None, :, :], dim=-1, eps=1e-8)
    f = cos if mode == 0 else (F.softmax(cos, 1) if mode == 1 else F.log_softmax(cos, 1))
    (f * go.double()).sum().backward()
    ref = pr.grad.float()
    assert torch.allclose(got, ref, atol=1e-5 + 2e-4 * float(ref.abs().max())), (got - ref).abs().max()


def test_weighted_feature_gradients(dev):
    """weighted_feature (RAGraph_node/downprompt.py:100-114): ELU(w0 a + w1 b) with the [1, 2] weight on the device -- the output
    and the gradients of a, b AND the weight against torch."""
    import torch.nn.functional as F
    from ragraph_amd import downprompt_node as dpn

    torch.manual_seed(5)
    a = torch.randn(37, 64, device=dev, requires_grad=True)
    b = torch.randn(37, 64, device=dev, requires_grad=True)
    mod = dpn.weighted_feature(2).to(dev)
    with torch.no_grad():
        mod.weight.copy_(torch.tensor([[0.7, -0.4]], device=dev))
    out = mod(a, b)
    go = torch.randn_like(out)
    (out * go).sum().backward()
    ar, br = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    wr = mod.weight.detach().clone().requires_grad_(True)
    ref = F.elu(wr[0][0] * ar + wr[0][1] * br)
    (ref * go).sum().backward()
    assert torch.allclose(out, ref, atol=1e-6)
    assert torch.allclose(a.grad, ar.grad, atol=1e-6) and torch.allclose(b.grad, br.grad, atol=1e-6)
    assert torch.allclose(mod.weight.grad, wr.grad, atol=1e-4), (mod.weight.grad, wr.grad)


@pytest.mark.parametrize("n,M,N", [(1, 1, 1), (7, 3, 5), (300, 3, 256), (1000, 256, 256), (4097, 130, 70), (100_000, 256, 256),
                                   (50_001, 3, 256)])
def test_linear_tn_matches_torch(dev, n, M, N):
    """gW = gY^T X without transposed copies (ragraph_linear_tn_f32): against torch in fp64 (1e-5 of the largest entry x sqrt
    of the rows summed), deterministic from call to call, and -- one row range -- the bits of ragraph_linear_f32 on transposed copies."""
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(n + M + N)
    a = torch.randn(n, M, device=dev, generator=g)
    b = torch.randn(n, N, device=dev, generator=g)
    got = K.linear_tn(a, b)
    ref = (a.double().t() @ b.double())
    assert got.shape == (M, N)
    assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())) * max(1.0, n ** 0.5 / 8)
    assert torch.equal(got, K.linear_tn(a, b))
    if n <= 256:   # a single range: the product through the dense kernel's chains over k = rows (transposed operands), bit for bit
        assert torch.equal(got, K.linear(a.t().contiguous(), b.t().contiguous()))


def test_column_sums_and_decoder_training_step_at_100k_rows(dev):
    """The bias gradient of 100 000 rows (one workgroup's 7.5-ms chain in round 4) as ranges summed chip-wide, and the decoder's
    training step on that many rows against torch autograd."""
    from ragraph_amd import autograd as A
    from ragraph_amd import kernels as K

    g = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn(100_000, 256, device=dev, generator=g)
    cs = K.column_sums(x)
    assert float((cs.double() - x.double().sum(0)).abs().max()) < 1e-2 and torch.equal(cs, K.column_sums(x))
    assert torch.equal(K.column_sums(x[:200]), K.segment_reduce(x[:200].contiguous(), torch.tensor([0, 200], device=dev)).reshape(-1))
    assert float((K.column_sums(x[:5000]).double() - x[:5000].double().sum(0)).abs().max()) < 1e-3     # ranges of 256 rows
    w = (0.05 * torch.randn(64, 256, device=dev, generator=g)).requires_grad_(True)
    bia = torch.zeros(64, device=dev, requires_grad=True)
    y = A.linear(x, w, bia, K.ACT_LEAKY, 0.01)
    tgt = torch.randn(100_000, 64, device=dev, generator=g)
    ((y - tgt) ** 2).mean().backward()
    w2, b2 = w.detach().clone().requires_grad_(True), bia.detach().clone().requires_grad_(True)
    y2 = torch.nn.functional.leaky_relu(torch.nn.functional.linear(x, w2, b2), 0.01)
    ((y2 - tgt) ** 2).mean().backward()
    assert close(w.grad, w2.grad) and close(bia.grad, b2.grad)
