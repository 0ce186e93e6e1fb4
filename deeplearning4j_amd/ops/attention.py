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


def _is_bthd(t):
    """[B, H, T, D] tensor that is a permuted view of a contiguous [B, T, H, D] buffer (the per-head view of a linear op's
    [B*T, H*D] output): the kernels read / write this layout directly, so no permute copy is made in either direction."""
    B, H, T, D = t.shape
    return t.stride() == (T * H * D, D, H * D, 1) and t.data_ptr() % 16 == 0 and H > 1


def _empty_like_layout(t, bthd):
    B, H, T, D = t.shape
    if bthd:
        return torch.empty(B, T, H, D, device=t.device, dtype=t.dtype).permute(0, 2, 1, 3)
    return torch.empty(B, H, T, D, device=t.device, dtype=t.dtype)


class _FlashSDPA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, scale, causal):
        from ..native import lib as L
        from . import linear as LIN
        B, H, Tq, D = q.shape
        Tk = k.shape[2]
        bthd = _is_bthd(q) and _is_bthd(k) and _is_bthd(v)
        qc, kc, vc = (q, k, v) if bthd else (q.contiguous(), k.contiguous(), v.contiguous())
        mk = None
        if mask is not None:
            mk = mask if (mask.dtype == torch.float32 and mask.is_contiguous()) else mask.float().contiguous()
        o = _empty_like_layout(qc, bthd)
        lse = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
        L.check(L.hip().dl4j_sdpa_fwd_l(L.ptr(qc), L.ptr(kc), L.ptr(vc), L.ptr(mk), L.ptr(o), L.ptr(lse), L.c_int(B), L.c_int(H),
                                        L.c_int(Tq), L.c_int(Tk), L.c_int(D), L.c_float(scale), L.c_int(1 if causal else 0),
                                        L.c_int(1 if bthd else 0), L.stream()), "sdpa_fwd")
        ctx.save_for_backward(qc, kc, vc, o, lse, mk)
        ctx.scale, ctx.causal, ctx.bthd = scale, causal, bthd
        return o

    @staticmethod
    def backward(ctx, do):
        """Fused backward (native/csrc/attention.hip: delta pre-pass, dQ pass, dK/dV pass; P is recomputed from the LSE)."""
        from ..native import lib as L
        q, k, v, o, lse, mk = ctx.saved_tensors
        B, H, Tq, D = q.shape
        Tk = k.shape[2]
        bthd = ctx.bthd
        dof = do if do.dtype == q.dtype else do.to(q.dtype)
        if bthd:
            if not _is_bthd(dof):
                dof = dof.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)
        else:
            dof = dof.contiguous()
        dq, dk, dv = _empty_like_layout(q, bthd), _empty_like_layout(k, bthd), _empty_like_layout(v, bthd)
        delta = torch.empty(B, H, Tq, device=q.device, dtype=torch.float32)
        L.check(L.hip().dl4j_sdpa_bwd_l(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mk), L.ptr(o), L.ptr(lse), L.ptr(dof), L.ptr(dq), L.ptr(dk),
                                        L.ptr(dv), L.ptr(delta), L.c_int(B), L.c_int(H), L.c_int(Tq), L.c_int(Tk), L.c_int(D),
                                        L.c_float(ctx.scale), L.c_int(1 if ctx.causal else 0), L.c_int(1 if bthd else 0), L.stream()),
                "sdpa_bwd")
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
