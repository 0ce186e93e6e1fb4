"""Row softmax family: fused softmax+MCXENT (score + gradient), softmax, log-softmax, softmax-bp.

GPU path: native/csrc/softmax_xent.hip. CPU / fp64 path: PyTorch (test oracle).
Reference semantics: nd4j-api/.../lossfunctions/impl/LossMCXENT.java:96-164 (softmax clip 1e-10).
"""
from __future__ import annotations

import torch

from ..native import lib as L


def _ok(z):
    return L.use_hip(z) and z.dtype in (torch.float32, torch.bfloat16) and z.is_contiguous() and z.dim() == 2


def softmax_xent(z, y=None, labels_idx=None, want_probs=False):
    """Returns (score_sum[1] fp32 tensor, grad = softmax(z)-y, probs or None).

    ``y`` one-hot/soft labels with z's shape, or ``labels_idx`` int64 class ids (sparse MCXENT).
    Score is the SUM over rows (caller divides by minibatch, as DL4J does).
    """
    M, N = z.shape
    if _ok(z) and (y is None or (y.dtype in (z.dtype, torch.float32) and y.is_contiguous())):
        # labels may stay fp32 (network dtype) while the logits are bf16: the kernel reads each operand in its own type
        grad = torch.empty_like(z)
        probs = torch.empty_like(z) if want_probs else None
        score = torch.empty(1, device=z.device, dtype=torch.float32)
        ycode = L.dtype_code(y) if y is not None else L.dtype_code(z)
        L.check(L.hip().dl4j_softmax_xent(L.ptr(z), L.ptr(y), L.ptr(labels_idx), L.ptr(probs), L.ptr(grad),
                                          L.ptr(None), L.ptr(score), L.c_int64(M), L.c_int(N),
                                          L.c_int(L.dtype_code(z) | (ycode << 8)), L.stream()), "softmax_xent")
        return score, grad, probs
    cd = torch.float32 if z.dtype in (torch.bfloat16, torch.float16) else z.dtype     # 1e-10 clip underflows in fp16
    p = torch.softmax(z.to(cd), dim=1)
    if y is None:
        yv = torch.zeros_like(p)
        yv.scatter_(1, labels_idx.view(-1, 1), 1.0)
    else:
        yv = y.to(cd)
    score = -(yv * torch.log(p.clamp(1e-10, 1.0))).sum().reshape(1)
    return score, (p - yv).to(z.dtype), (p.to(z.dtype) if want_probs else None)


def softmax(z, log=False):
    if _ok(z):
        out = torch.empty_like(z)
        L.check(L.hip().dl4j_softmax_rows(L.ptr(z), L.ptr(out), L.c_int64(z.shape[0]), L.c_int(z.shape[1]),
                                          L.c_int(1 if log else 0), L.c_int(L.dtype_code(z)), L.stream()), "softmax")
        return out
    return torch.log_softmax(z, dim=-1) if log else torch.softmax(z, dim=-1)


def softmax_bp(p, dy):
    if _ok(p) and dy.is_contiguous() and dy.dtype == p.dtype:
        dz = torch.empty_like(p)
        L.check(L.hip().dl4j_softmax_bwd(L.ptr(p), L.ptr(dy), L.ptr(dz), L.c_int64(p.shape[0]), L.c_int(p.shape[1]),
                                         L.c_int(L.dtype_code(p)), L.stream()), "softmax_bwd")
        return dz
    return p * (dy - (dy * p).sum(-1, keepdim=True))
