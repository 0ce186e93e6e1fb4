"""2-D pooling (max / avg / pnorm) forward + backward.

GPU channels-last fp32/bf16: native/csrc/pooling.hip (argmax-byte forward, atomic-free gather
backward). Everything else (CPU, fp64, NCHW-contiguous, pnorm): PyTorch composite.
Argument contract mirrors libnd4j maxpool2d/avgpool2d iArgs
[kH,kW,sH,sW,pH,pW,dH,dW,same,extra,format] (SubsamplingLayer.java:139-143).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F

from ..native import lib as L


def out_size(i, k, s, p, d=1):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


def _is_cl(x):
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)


def _geom(x, OH, OW, k, s, p, d, excl):
    N, C, H, W = x.shape
    return (ctypes.c_int * 15)(N, H, W, C, OH, OW, k[0], k[1], s[0], s[1], p[0], p[1], d[0], d[1], 1 if excl else 0)


class PoolCtx:
    __slots__ = ("amax", "x", "y", "geom", "mode", "shape", "args")


def pool2d_forward(x, mode, k, s, p, d=(1, 1), avg_exclude_pad=False, pnorm=2, train=True):
    """x: logical NCHW tensor. Returns (y, ctx)."""
    N, C, H, W = x.shape
    OH, OW = out_size(H, k[0], s[0], p[0], d[0]), out_size(W, k[1], s[1], p[1], d[1])
    ctx = PoolCtx()
    ctx.mode, ctx.shape, ctx.args = mode, x.shape, (k, s, p, d, avg_exclude_pad, pnorm)
    if (mode in ("max", "avg") and L.use_hip(x) and x.dtype in (torch.float32, torch.bfloat16) and _is_cl(x)
            and k[0] * k[1] <= 255):
        y = torch.empty((N, C, OH, OW), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        amax = None
        if mode == "max" and train:
            amax = torch.empty((N, OH, OW, C), device=x.device, dtype=torch.uint8)
        g = _geom(x, OH, OW, k, s, p, d, avg_exclude_pad)
        L.check(L.hip().dl4j_pool2d_fwd(L.ptr(x), L.ptr(y), L.ptr(amax), g, L.c_int(0 if mode == "max" else 1),
                                        L.c_int(L.dtype_code(x)), L.stream()), "pool2d_fwd")
        ctx.amax, ctx.geom, ctx.x, ctx.y = amax, g, None, None
        return y, ctx
    # composite path
    ctx.amax = None
    if mode == "max":
        y, idx = F.max_pool2d(x, k, s, p, d, return_indices=True)
        ctx.x, ctx.y = idx, None
    elif mode == "avg":
        y = F.avg_pool2d(x, k, s, p, count_include_pad=not avg_exclude_pad)
        ctx.x = None
    elif mode == "pnorm":
        xp = x.abs().pow(pnorm)
        y = (F.avg_pool2d(xp, k, s, p, count_include_pad=True) * (k[0] * k[1])).pow(1.0 / pnorm)
        ctx.x, ctx.y = x, y
    else:
        raise ValueError(mode)
    return y, ctx


def pool2d_backward(dy, ctx):
    k, s, p, d, excl, pnorm = ctx.args
    N, C, H, W = ctx.shape
    if ctx.mode in ("max", "avg") and getattr(ctx, "geom", None) is not None and (ctx.amax is not None or ctx.mode == "avg"):
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=dy.dtype, memory_format=torch.channels_last)
        L.check(L.hip().dl4j_pool2d_bwd(L.ptr(dy), L.ptr(ctx.amax), L.ptr(dx), ctx.geom,
                                        L.c_int(0 if ctx.mode == "max" else 1), L.c_int(L.dtype_code(dy)),
                                        L.stream()), "pool2d_bwd")
        return dx
    if ctx.mode == "max":
        idx = ctx.x
        dx = torch.zeros((N, C, H * W), device=dy.device, dtype=dy.dtype)
        dx.scatter_add_(2, idx.reshape(N, C, -1), dy.reshape(N, C, -1))
        return dx.view(N, C, H, W)
    if ctx.mode == "avg":
        x0 = torch.zeros((N, C, H, W), device=dy.device, dtype=dy.dtype, requires_grad=True)
        with torch.enable_grad():
            y = F.avg_pool2d(x0, k, s, p, count_include_pad=not excl)
        return torch.autograd.grad(y, x0, dy)[0]
    # pnorm: y = (sum |x|^p)^(1/p)  =>  dx = sign(x)|x|^(p-1) * scatter(dy * y^(1-p)); all-zero windows contribute 0
    # (autograd through pow(1/p) gives 0 * inf = NaN there, e.g. after a ReLU)
    x, y = ctx.x.detach(), ctx.y.detach()
    gy = torch.where(y > 0, dy * y.clamp_min(1e-30).pow(1.0 - pnorm), torch.zeros_like(dy))
    leaf = torch.zeros_like(x, requires_grad=True)
    with torch.enable_grad():
        lin = F.avg_pool2d(leaf, k, s, p, count_include_pad=True) * (k[0] * k[1])
    spread = torch.autograd.grad(lin, leaf, gy)[0]
    return spread * x.sign() * x.abs().pow(pnorm - 1)
