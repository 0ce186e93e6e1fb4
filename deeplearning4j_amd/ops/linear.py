"""Differentiable y = x . W (+ b) for SameDiff / op-level matmul on the in-tree MFMA kernels (forward AND backward), so no
rocBLAS / hipBLASLt kernel appears in an imported BERT step:

    forward   y  = gemm_nt(x, W^T, bias)                 W [K, N] as declared by xw_plus_b / matmul; W^T is a small copy
    backward  dx = gemm_nt(dy, W)                        W [K, N] IS the "NT" B operand of this product: no copy
              dW = (split-K weight-gradient kernel: dy^T . x -> [N, K] fp32)^T
              db = column sums of dy

Reference: ops/declarable/generic/nn/xw_plus_b.cpp (+ xw_plus_b_bp), generic/blas/matmul.cpp, helpers/cuda_off/MmulHelper.cu.
"""
from __future__ import annotations

import torch

from . import gemm as G
from . import dense as DENSE


def usable(x, w):
    """x [..., K], w [K, N] on the GPU, one of bf16 / fp16 / fp32, K and N multiples of the 16-byte chunk."""
    if not (isinstance(x, torch.Tensor) and isinstance(w, torch.Tensor) and x.is_cuda and w.is_cuda and x.dtype == w.dtype
            and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and w.dim() == 2 and x.dim() >= 2):
        return False
    ch = G.chunk(x.dtype)
    K, N = w.shape
    return x.shape[-1] == K and K % ch == 0 and N % ch == 0 and x.numel() // K >= 16


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        wt = w.t().contiguous()
        b32 = None if bias is None else bias.reshape(-1).float()
        y = G.gemm_nt(x2, wt, bias=b32)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.bias_shape = None if bias is None else bias.shape
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], w.shape[1])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        d2 = dy.reshape(-1, dy.shape[-1])
        if not d2.is_contiguous() or d2.dtype != x2.dtype:
            d2 = d2.to(x2.dtype).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wc = w if w.is_contiguous() else w.contiguous()
            dx = G.gemm_nt(d2, wc).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            K, N = w.shape
            dwt = torch.zeros(N, K, device=w.device, dtype=torch.float32)
            if DENSE.dw_ok(d2, x2, dwt):
                DENSE.dw(d2, x2, dwt, True)
                dw = dwt.t().to(w.dtype)
            else:
                dw = (x2.float().t() @ d2.float()).to(w.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = d2.sum(0, dtype=torch.float32).to(ctx.bias_dtype).reshape(ctx.bias_shape)
        return dx, dw, db


def linear(x, w, bias=None):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w, bias)):
        return LinearFn.apply(x, w, bias)
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = G.gemm_nt(x2, w.t().contiguous(), bias=None if bias is None else bias.reshape(-1).float())
    return y.view(*x.shape[:-1], w.shape[1])
