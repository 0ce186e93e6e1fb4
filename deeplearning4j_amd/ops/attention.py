"""Scaled dot-product attention (dot_product_attention_v2 semantics: softmax(QK^T*scale)V).

Reference: libnd4j/include/ops/declarable/generic/nn/dot_product_attention_v2.cpp:35-181 (materialises
the full [b,h,Tq,Tk] score matrix and returns it). On the GPU (bf16, head dim 64/128, key-padding mask or none, no
attention dropout) the forward is the hand-written flash-style kernel native/csrc/attention.hip, which never materialises
scores; the backward is fused too (delta pre-pass + a dQ pass + a dK/dV pass that recompute the probabilities from the
saved log-sum-exp). ``return_scores=True`` reproduces the reference outputs; other cases use ATen's SDPA.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _hip_ok(q, k, v, mask, dropout_p, training):
    from ..native import lib as L
    if not (L.use_hip(q, k, v) and q.dtype == k.dtype == v.dtype == torch.bfloat16 and q.dim() == 4):
        return False
    if q.shape[-1] not in (64, 128) or k.shape[-1] != q.shape[-1] or v.shape[-1] != q.shape[-1] or k.shape[2] != v.shape[2]:
        return False
    if q.shape[:2] != k.shape[:2] or q.shape[:2] != v.shape[:2] or (dropout_p and training):
        return False
    return mask is None or (mask.dim() == 2 and mask.shape == (q.shape[0], k.shape[2]))


class _FlashSDPA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, scale, causal):
        from ..native import lib as L
        B, H, Tq, D = q.shape
        Tk = k.shape[2]
        qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
        mk = mask.float().contiguous() if mask is not None else None
        o = torch.empty_like(qc)
        lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
        L.check(L.hip().dl4j_sdpa_fwd(L.ptr(qc), L.ptr(kc), L.ptr(vc), L.ptr(mk), L.ptr(o), L.ptr(lse), L.c_int(B), L.c_int(H),
                                      L.c_int(Tq), L.c_int(Tk), L.c_int(D), L.c_float(scale), L.c_int(1 if causal else 0), L.stream()),
                "sdpa_fwd")
        ctx.save_for_backward(qc, kc, vc, o, lse, mk)
        ctx.scale, ctx.causal = scale, causal
        return o

    @staticmethod
    def backward(ctx, do):
        """Fused backward (native/csrc/attention.hip: delta pre-pass, dQ pass, dK/dV pass; P is recomputed from the LSE)."""
        from ..native import lib as L
        q, k, v, o, lse, mk = ctx.saved_tensors
        B, H, Tq, D = q.shape
        Tk = k.shape[2]
        dof = do.to(q.dtype).contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
        L.check(L.hip().dl4j_sdpa_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mk), L.ptr(o), L.ptr(lse), L.ptr(dof), L.ptr(dq), L.ptr(dk),
                                      L.ptr(dv), L.ptr(delta), L.c_int(B), L.c_int(H), L.c_int(Tq), L.c_int(Tk), L.c_int(D),
                                      L.c_float(ctx.scale), L.c_int(1 if ctx.causal else 0), L.stream()), "sdpa_bwd")
        return dq, dk, dv, None, None, None


def sdpa_backward_reference(q, k, v, o, lse, mk, do, scale, causal):
    """The same backward with library GEMMs and a materialised probability matrix (test oracle for the fused kernels)."""
    s = torch.matmul(q, k.transpose(-1, -2)).float() * scale
    if mk is not None:
        s = s.masked_fill(mk[:, None, None, :] == 0, float("-inf"))
    if causal:
        Tq, Tk = s.shape[-2:]
        s = s.masked_fill(~torch.ones(Tq, Tk, dtype=torch.bool, device=s.device).tril(), float("-inf"))
    p = torch.exp(s - lse.unsqueeze(-1))                      # fully masked rows: lse = +inf -> p = 0
    dof = do.contiguous()
    pb = p.to(q.dtype)
    dv = torch.matmul(pb.transpose(-1, -2), dof)
    dp = torch.matmul(dof, v.transpose(-1, -2)).float()
    delta = (dof.float() * o.float()).sum(-1, keepdim=True)
    ds = (p * (dp - delta) * scale).to(q.dtype)
    return torch.matmul(ds, k), torch.matmul(ds.transpose(-1, -2), q), dv


def sdpa(q, k, v, mask=None, scale=None, causal=False, dropout_p=0.0, training=False, return_scores=False):
    """q,k,v: [b, h, T, d]. mask: [b, Tk] (1 = keep) or broadcastable to [b,h,Tq,Tk]."""
    d = q.shape[-1]
    scale = (1.0 / math.sqrt(d)) if not scale else scale
    if not return_scores and _hip_ok(q, k, v, mask, dropout_p, training):
        return _FlashSDPA.apply(q, k, v, mask, float(scale), bool(causal))
    am = None
    if mask is not None:
        m = mask
        if m.dim() == 2:
            m = m.reshape(m.shape[0], 1, 1, m.shape[1])
        am = torch.zeros_like(m, dtype=q.dtype).masked_fill(m == 0, float("-inf"))
    if not return_scores:
        return F.scaled_dot_product_attention(q, k, v, attn_mask=am, dropout_p=dropout_p if training else 0.0,
                                              is_causal=causal and am is None, scale=scale)
    logits = torch.matmul(q, k.transpose(-1, -2)) * scale
    if am is not None:
        logits = logits + am
    if causal:
        Tq, Tk = logits.shape[-2:]
        cm = torch.ones(Tq, Tk, dtype=torch.bool, device=q.device).tril()
        logits = logits.masked_fill(~cm, float("-inf"))
    scores = torch.softmax(logits, dim=-1)
    p = F.dropout(scores, dropout_p, training) if dropout_p and training else scores
    return torch.matmul(p, v), scores, logits
