"""Autograd wrappers of the trainable parts of the path.

  * node / graph flavours: the encoder output is detached (preprompt.py:62) and the bank carries no gradient, so
    fine-tuning (finetune-rag.py:81-84) back-propagates through the decoder head only (RAGraph.py:53-57): linear,
    axpby, softmax-mix.
  * few-shot flavours train the SECOND GCN layer through decode() (RAGraph_node_fewshot/RAGraph.py:69,
    RAGraph_graph_fewshot/RAGraph.py:77): spmm_csr (backward = the SpMM over the transposed CSR) with its fused
    bias / PReLU epilogue, slope and bias gradients included.
  * edge flavour (RAGraph_edge/modules/RAGraph.py:265-355): embeddings, gate and LoRA factors through the gate, three
    propagation layers and the batch gathers: sigmoid_gate, spmm_csr, gather_rows.
Every forward AND backward product runs on the HIP kernels (the matrix parts of a backward pass are the forward entry
points on transposed operands; the element-wise derivatives are csrc/rowops.hip's *_grad kernels).
"""
from __future__ import annotations

import torch

from . import kernels as K


class _Linear(torch.autograd.Function):
    """y = act(x @ W^T + b), act in {none, leaky(alpha)}."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, alpha):
        y = K.linear(x, weight, bias, act=act, alpha=alpha)
        ctx.save_for_backward(x, weight, y)
        ctx.act, ctx.alpha, ctx.has_bias = act, alpha, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.act != K.ACT_NONE:
            gy = K.act_grad(y, gy, ctx.act, ctx.alpha)  # sign(y) == sign(pre-activation) for alpha > 0
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = K.linear(gy, w.t().contiguous())                      # gy @ W
        if ctx.needs_input_grad[1]:
            if gy.shape[0] >= K.LINEAR_TN_MIN_ROWS:
                gw = K.linear_tn(gy, x)                                # gy^T @ x, no transposed copies (16-graph batches: 0.84 ms
            else:                                                      # a step either way; 100 000 rows: 1.8 -> 0.2 ms a product)
                gw = K.linear(gy.t().contiguous(), x.t().contiguous())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = K.column_sums(gy)
        return gx, gw, gb, None, None


def linear(x, weight, bias=None, act=K.ACT_NONE, alpha=0.0):
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _Linear.apply(x, weight, bias, act, alpha)
    return K.linear(x, weight, bias, act=act, alpha=alpha)


class _SoftmaxMix(torch.autograd.Function):
    """softmax(logits)*(1-lam) + rag_label*lam (RAGraph.py:55-57)."""

    @staticmethod
    def forward(ctx, logits, rag_label, lam):
        # the plain softmax is what backward needs: computed once and mixed by the axpby kernel -- the same two
        # multiplies and one add as the fused kernel, so training and inference outputs agree bit for bit (and
        # label_weight = 1 needs no division by 1 - lam)
        p = K.softmax_mix(logits, None)
        ctx.save_for_backward(p)
        ctx.lam = lam if rag_label is not None else 0.0
        return p if rag_label is None else K.axpby(p, 1.0 - lam, rag_label, lam)

    @staticmethod
    def backward(ctx, go):
        (p,) = ctx.saved_tensors
        return K.softmax_grad(p, go.contiguous(), 1.0 - ctx.lam), None, None


def softmax_mix(logits, rag_label, lam):
    if torch.is_grad_enabled() and logits.requires_grad:
        return _SoftmaxMix.apply(logits, rag_label, lam)
    return K.softmax_mix(logits, rag_label, lam)


class _Axpby(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, wa, b, wb):
        ctx.wa, ctx.wb = wa, wb
        return K.axpby(a, wa, b, wb)

    @staticmethod
    def backward(ctx, go):
        go = go.contiguous()
        zero = 0.0
        return (K.axpby(go, ctx.wa, go, zero) if ctx.needs_input_grad[0] else None, None,
                K.axpby(go, ctx.wb, go, zero) if ctx.needs_input_grad[2] else None, None)


def axpby(a, wa, b, wb):
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return _Axpby.apply(a, wa, b, wb)
    return K.axpby(a, wa, b, wb)


class _SpmmCsr(torch.autograd.Function):
    """y = act(A @ x + bias) for a CSRGraph A -- layers/gcn.py:36-40, Propagation.py:22-25, edge _agg (:232-240).
    Backward: gz = gy * act'(z); gx = A^T gz (the same kernel over the transposed CSR, cached on the graph); gbias =
    column sums of gz; PReLU slope: sum of gy * z over z < 0.  The fused epilogue keeps only y, from which the
    pre-activation's sign and value follow while the slope is positive (sign(y) = sign(z), z = y / alpha on the negative
    side).  A trained PReLU slope is unconstrained: for alpha <= 0 the forward runs unfused and keeps z itself."""

    @staticmethod
    def forward(ctx, g, x, bias, act, alpha_t, alpha):
        ctx.keeps_z = act == K.ACT_PRELU and alpha <= 0.0
        if ctx.keeps_z:
            z = K.spmm_csr(g.rowptr, g.col, g.val, x, bias=bias, act=K.ACT_NONE, long_rows=g.has_long_rows)
            y = K.mul_cols(z, torch.ones(z.shape[-1], device=z.device), act, alpha)    # act(z * 1)
            ctx.save_for_backward(z)
        else:
            y = K.spmm_csr(g.rowptr, g.col, g.val, x, bias=bias, act=act, alpha=alpha, long_rows=g.has_long_rows)
            ctx.save_for_backward(y)
        ctx.g, ctx.act, ctx.alpha = g, act, alpha
        ctx.has_bias, ctx.has_alpha = bias is not None, alpha_t is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors          # (z itself when keeps_z: act_grad's sign test is then on the pre-activation)
        gy = gy.contiguous()
        galpha = None
        if ctx.act != K.ACT_NONE:
            want_alpha = ctx.has_alpha and ctx.needs_input_grad[4]
            if ctx.keeps_z:
                gz = K.act_grad(y, gy, ctx.act, ctx.alpha)
                if want_alpha:             # sum of gy * min(z, 0): min(z, 0) = z - relu(z)
                    neg = K.axpby(y, 1.0, K.mul_cols(y, torch.ones(y.shape[-1], device=y.device), K.ACT_RELU), -1.0)
                    terms = K.mul(gy, neg)
            elif want_alpha:
                gz, terms = K.act_grad(y, gy, ctx.act, ctx.alpha, want_alpha_terms=True)
            else:
                gz = K.act_grad(y, gy, ctx.act, ctx.alpha)
            if want_alpha:
                galpha = K.column_sums(terms).sum().reshape(1)   # (the last 256 -> 1 is bookkeeping)
        else:
            gz = gy
        gx = gb = None
        if ctx.needs_input_grad[1]:
            gt = ctx.g.transposed()
            gx = K.spmm_csr(gt.rowptr, gt.col, gt.val, gz, long_rows=gt.has_long_rows)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = K.column_sums(gz)
        return None, gx, gb, None, galpha, None


def spmm_csr(g, x, bias=None, act=K.ACT_NONE, alpha_param=None, alpha=0.0):
    """Differentiable act(A @ x + bias); alpha_param = the PReLU weight tensor (its gradient is produced when it requires
    one), alpha = its value as a host scalar."""
    need = torch.is_grad_enabled() and (x.requires_grad or (bias is not None and bias.requires_grad) or
                                        (alpha_param is not None and alpha_param.requires_grad))
    if need:
        return _SpmmCsr.apply(g, x, bias, act, alpha_param, alpha)
    return K.spmm_csr(g.rowptr, g.col, g.val, x, bias=bias, act=act, alpha=alpha, long_rows=g.has_long_rows)


class _SigmoidGate(torch.autograd.Function):
    """x * sigmoid(z) -- RAGraph_edge/modules/RAGraph.py:168."""

    @staticmethod
    def forward(ctx, x, z):
        ctx.save_for_backward(x, z)
        return K.sigmoid_gate(x, z)

    @staticmethod
    def backward(ctx, g):
        x, z = ctx.saved_tensors
        return K.sigmoid_gate_grad(x, z, g.contiguous())


def sigmoid_gate(x, z):
    if torch.is_grad_enabled() and (x.requires_grad or z.requires_grad):
        return _SigmoidGate.apply(x, z)
    return K.sigmoid_gate(x, z)


class _GatherRows(torch.autograd.Function):
    """v[idx] for a 1-D index -- batch_user_emb = user_emb[users] (modules/RAGraph.py:343-345).  Backward scatters the
    row gradients back: rows hit several times are summed by a CSR SpMM (rows = bank rows, one unit entry per hit),
    deterministic -- no atomics."""

    @staticmethod
    def forward(ctx, v, idx):
        ctx.n = v.shape[0]
        ctx.save_for_backward(idx)
        return K.gather_rows(v, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        from .graph import CSRGraph
        m = idx.numel()
        hits, _ = CSRGraph.from_coo(idx.reshape(-1), torch.arange(m, device=idx.device), torch.ones(m, device=idx.device), ctx.n)
        hits.n_cols = m
        return K.spmm_csr(hits.rowptr, hits.col, hits.val, g.contiguous().reshape(m, -1)), None


def gather_rows(v, idx):
    if torch.is_grad_enabled() and v.requires_grad:
        return _GatherRows.apply(v, idx)
    return K.gather_rows(v, idx)


class _MulCols(torch.autograd.Function):
    """y = act(x * w), w one weight per column -- downstreamprompt.forward (RAGraph_graph/downprompt.py:164-168; ELU in
    RAGraph_node/downprompt.py:118-130).  gz = gy * act'(z) through the output (ELU: y > 0 ? 1 : y + alpha);
    gx = gz * w; gw = column sums of gz * x, rows added in order (segment_reduce: deterministic)."""

    @staticmethod
    def forward(ctx, x, w, act, alpha):
        y = K.mul_cols(x, w, act, alpha)
        ctx.save_for_backward(x, w, y)
        ctx.act, ctx.alpha = act, alpha
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gz = gy.contiguous()
        if ctx.act != K.ACT_NONE:
            gz = K.act_grad(y, gz, ctx.act, ctx.alpha)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = K.mul_cols(gz, w)
        if ctx.needs_input_grad[1]:
            D = x.shape[-1]
            prod = K.mul(gz.reshape(-1, D), x.reshape(-1, D))
            gw = K.column_sums(prod).reshape(w.shape)
        return gx, gw, None, None


def mul_cols(x, w, act=K.ACT_NONE, alpha=0.0):
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        return _MulCols.apply(x, w, act, alpha)
    return K.mul_cols(x, w, act, alpha)



class _ProtoCosine(torch.autograd.Function):
    """f(cosine(emb_g, proto_c)), f = identity / softmax / log_softmax (downprompt.py:41-56 graph flavour,
    RAGraph_node/downprompt.py:41-46).  The gradient goes to the embeddings and -- when a training step keeps the prototypes
    in the graph (RAGraph_node/downprompt.py:24-25: averageemb of the step's own embeddings) -- to the prototypes."""

    @staticmethod
    def forward(ctx, emb, proto, mode):
        out = K.proto_cosine(emb, proto, mode)
        ctx.save_for_backward(emb, proto, out)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, go):
        emb, proto, out = ctx.saved_tensors
        go = go.contiguous()
        gemb = K.proto_cosine_grad(emb, proto, ctx.mode, out, go) if ctx.needs_input_grad[0] else None
        gproto = K.proto_cosine_grad_proto(emb, proto, ctx.mode, out, go) if ctx.needs_input_grad[1] else None
        return gemb, gproto, None


def proto_cosine(emb, proto, mode=0):
    if torch.is_grad_enabled() and (emb.requires_grad or proto.requires_grad):
        return _ProtoCosine.apply(emb, proto, mode)
    return K.proto_cosine(emb, proto, mode)


class _SegmentSum(torch.autograd.Function):
    """Per-segment sum of rows (K.segment_reduce, rows added in index order); backward: every row takes its segment's
    gradient row (a gather)."""

    @staticmethod
    def forward(ctx, x, seg_ptr):
        ctx.save_for_backward(seg_ptr)
        ctx.n = x.shape[0]
        return K.segment_reduce(x, seg_ptr)

    @staticmethod
    def backward(ctx, g):
        (seg_ptr,) = ctx.saved_tensors
        counts = seg_ptr[1:] - seg_ptr[:-1]
        seg_of_row = torch.repeat_interleave(torch.arange(counts.numel(), device=g.device), counts, output_size=ctx.n)
        return K.gather_rows(g.contiguous(), seg_of_row), None


def segment_sum(x, seg_ptr):
    """Rows [seg_ptr[s], seg_ptr[s + 1]) of x summed per segment; seg_ptr must cover all rows of x."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _SegmentSum.apply(x, seg_ptr)
    return K.segment_reduce(x, seg_ptr)


class _Mix2(torch.autograd.Function):
    """a * w[0] + b * w[1] for a trainable [1, 2] weight read on the device (weighted_feature, RAGraph_node/downprompt.py:
    100-114).  ga = go * w[0], gb = go * w[1], gw = (sum go * a, sum go * b) -- sums in a fixed order (segment_reduce)."""

    @staticmethod
    def forward(ctx, a, b, w):
        ctx.save_for_backward(a, b, w)
        return K.axpby_dev(a, b, w, 0, 1)

    @staticmethod
    def backward(ctx, go):
        a, b, w = ctx.saved_tensors
        go = go.contiguous()
        ga = K.axpby_dev(go, go, w, 0, -1) if ctx.needs_input_grad[0] else None
        gb = K.axpby_dev(go, go, w, 1, -1) if ctx.needs_input_grad[1] else None
        gw = None
        if ctx.needs_input_grad[2]:
            def total(t):  # all elements of t added in a fixed order: columns of the row sums, then those
                D = t.shape[-1]
                rows = t.reshape(-1, D)
                col = K.column_sums(rows)
                return K.segment_reduce(col.reshape(D, 1).contiguous(), torch.tensor([0, D], dtype=torch.int64, device=t.device))
            gw = torch.cat([total(K.mul(go, a)), total(K.mul(go, b))], 1).reshape(w.shape)
        return ga, gb, gw


def mix2(a, b, w):
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad or w.requires_grad):
        return _Mix2.apply(a, b, w)
    return K.axpby_dev(a, b, w, 0, 1)
