"""Scaled dot-product attention (dot_product_attention_v2 semantics: softmax(QK^T*scale)V).

Reference: libnd4j/include/ops/declarable/generic/nn/dot_product_attention_v2.cpp:35-181 (materialises
the full [b,h,Tq,Tk] score matrix and returns it). Here the default path is the fused flash-style
kernel that never materialises scores; ``return_scores=True`` reproduces the reference outputs.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def sdpa(q, k, v, mask=None, scale=None, causal=False, dropout_p=0.0, training=False, return_scores=False):
    """q,k,v: [b, h, T, d]. mask: [b, Tk] (1 = keep) or broadcastable to [b,h,Tq,Tk]."""
    d = q.shape[-1]
    scale = (1.0 / math.sqrt(d)) if not scale else scale
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
