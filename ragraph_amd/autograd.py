"""Autograd wrappers for the only trainable part of the path: the decoder head (TaskDecoder, fusion, softmax-mix).

In the reference the encoder output is detached (preprompt.py:62) and the bank carries no gradient, so fine-tuning
(finetune-rag.py:81-84) back-propagates through RAGraph.py:53-57 only.  Forward AND backward GEMMs run on the HIP
linear kernel; the element-wise derivative masks are plain tensor ops on the device (training-side bookkeeping).
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
        if ctx.act in (K.ACT_LEAKY, K.ACT_PRELU):
            gy = torch.where(y >= 0, gy, gy * ctx.alpha)  # sign(y) == sign(pre-activation) for alpha > 0
        elif ctx.act == K.ACT_RELU:
            gy = torch.where(y > 0, gy, torch.zeros_like(gy))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = K.linear(gy, w.t().contiguous())                      # gy @ W
        if ctx.needs_input_grad[1]:
            gw = K.linear(gy.t().contiguous(), x.t().contiguous())     # gy^T @ x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            seg = torch.tensor([0, gy.shape[0]], dtype=torch.int64, device=gy.device)
            if gy.shape[1] % 4 == 0:
                gb = K.segment_reduce(gy, seg).reshape(-1)
            else:
                gb = gy.sum(0)
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
        g = go * (1.0 - ctx.lam)
        return p * (g - (g * p).sum(dim=-1, keepdim=True)), None, None


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
        return (go * ctx.wa if ctx.needs_input_grad[0] else None, None,
                go * ctx.wb if ctx.needs_input_grad[2] else None, None)


def axpby(a, wa, b, wb):
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return _Axpby.apply(a, wa, b, wb)
    return K.axpby(a, wa, b, wb)
