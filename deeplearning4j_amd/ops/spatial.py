"""Autograd entry points of the spatial helper kernels (native/csrc/helpers_spatial.hip, helpers_nn.hip LRN): depthwise conv2d,
3-D pooling, nearest 3-D upsampling and cross-channel LRN. On a GPU tensor the HIP kernels run in both directions; on the
CPU (tests without a GPU) the same functions fall back to ATen so the layers have one code path.

Reference layers: nn/layers/convolution/{DepthwiseConvolution2DLayer,SeparableConvolution2DLayer}.java,
nn/layers/convolution/subsampling/Subsampling3DLayer.java, nn/layers/convolution/upsampling/Upsampling3D.java,
nn/layers/normalization/LocalResponseNormalization.java (their CUDA side: libnd4j helpers/cuda/convolutions_depthwiseConv2d*.cu,
convolutions_pooling3d*.cu, convolutions_upsampling3d*.cu, lrn.cu).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import helpers as H


def _hip(*ts):
    return all(t is None or (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16)) for t in ts) and ts[0].is_cuda


class _Depthwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, wk, bias, s, p_lo, dl, out_hw):
        x = x.contiguous()
        wk = wk.to(x.dtype).contiguous()
        ctx.save_for_backward(x, wk)
        ctx.geom = (s, p_lo, dl, bias is not None, None if bias is None else bias.dtype)
        return H.depthwise_fwd(x, wk, bias, s, p_lo, dl, out_hw)

    @staticmethod
    def backward(ctx, gy):
        x, wk = ctx.saved_tensors
        s, p_lo, dl, has_b, bdt = ctx.geom
        dx, dw, db = H.depthwise_bwd(x, wk, gy.to(x.dtype).contiguous(), s, p_lo, dl, want_bias=has_b)
        return dx, dw.to(wk.dtype), (db.to(bdt) if has_b else None), None, None, None, None


def depthwise_conv2d(x, wk, bias, stride, pads, dilation):
    """x [N, C, H, W]; wk [C*mC, kH, kW]; pads = ((top, bottom), (left, right)). Differentiable in x, wk, bias."""
    s, dl = (int(stride[0]), int(stride[1])), (int(dilation[0]), int(dilation[1]))
    (pt, pb), (pl, pr) = pads
    kH, kW = wk.shape[1], wk.shape[2]
    oH = (x.shape[2] + pt + pb - dl[0] * (kH - 1) - 1) // s[0] + 1
    oW = (x.shape[3] + pl + pr - dl[1] * (kW - 1) - 1) // s[1] + 1
    if _hip(x, wk):
        return _Depthwise.apply(x, wk, bias, s, (int(pt), int(pl)), dl, (oH, oW))
    xp = F.pad(x, [pl, pr, pt, pb]) if (pt != pb or pl != pr) else x
    pad = (0, 0) if (pt != pb or pl != pr) else (pt, pl)
    return F.conv2d(xp, wk.unsqueeze(1).to(x.dtype), None if bias is None else bias.to(x.dtype), s, pad, dl, groups=x.shape[1])


class _Pool3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p_lo, dl, mode, include_pad, out_dhw):
        x = x.contiguous()
        y = H.pool3d(x, k, s, p_lo, dl, mode, include_pad, out_dhw=out_dhw)
        ctx.save_for_backward(x, y)
        ctx.geom = (k, s, p_lo, dl, mode, include_pad)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y = ctx.saved_tensors
        k, s, p_lo, dl, mode, include_pad = ctx.geom
        return H.pool3d_bwd(x, y, gy.to(x.dtype).contiguous(), k, s, p_lo, dl, mode, include_pad), None, None, None, None, None, None, None


def pool3d(x, kernel, stride, pads, dilation, mode, include_pad=True):
    """x [N, C, D, H, W]; pads = ((front, back), (top, bottom), (left, right)); mode 'max' | 'avg'. Padding never enters a max;
    an average divides by the full window (include_pad) or by the in-bounds count."""
    k, s, dl = [int(v) for v in kernel], [int(v) for v in stride], [int(v) for v in dilation]
    out = [(x.shape[2 + a] + pads[a][0] + pads[a][1] - dl[a] * (k[a] - 1) - 1) // s[a] + 1 for a in range(3)]
    if _hip(x):
        return _Pool3d.apply(x, k, s, [int(p[0]) for p in pads], dl, mode, bool(include_pad), out)
    flat = []
    for lo, hi in reversed(pads):
        flat += [int(lo), int(hi)]
    if mode == "max":
        return F.max_pool3d(F.pad(x, flat, value=float("-inf")) if any(flat) else x, k, s, 0, dl)
    xs = F.avg_pool3d(F.pad(x, flat) if any(flat) else x, k, s, 0, False, True, divisor_override=1)
    if include_pad:
        return xs / float(k[0] * k[1] * k[2])
    ones = F.pad(torch.ones((1, 1) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device), flat)
    return xs / F.avg_pool3d(ones, k, s, 0, False, True, divisor_override=1).clamp_min(1.0)


class _Upsample3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f):
        ctx.f = f
        return H.upsample3d(x.contiguous(), *f)

    @staticmethod
    def backward(ctx, gy):
        return H.upsample3d_bwd(gy.contiguous(), *ctx.f), None


def upsample3d(x, factors):
    f = tuple(int(v) for v in factors)
    if _hip(x):
        return _Upsample3d.apply(x, f)
    for a, r in enumerate(f):
        x = x.repeat_interleave(r, dim=2 + a)
    return x


class _LRN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, depth, bias, alpha, beta):
        xl = x.permute(0, 2, 3, 1).contiguous()
        ctx.save_for_backward(xl)
        ctx.args = (depth, bias, alpha, beta)
        return H.lrn_last(xl, depth, bias, alpha, beta).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (xl,) = ctx.saved_tensors
        depth, bias, alpha, beta = ctx.args
        gl = gy.permute(0, 2, 3, 1).contiguous().to(xl.dtype)
        return H.lrn_bwd_last(xl, gl, depth, bias, alpha, beta).permute(0, 3, 1, 2), None, None, None, None


def lrn(x, n, k, alpha, beta):
    """Cross-channel LRN on NCHW: y = x / (k + alpha * sum_{window n} x^2)^beta."""
    n = int(n)
    if _hip(x) and x.dim() == 4 and n % 2 == 1 and x.shape[1] <= 4096:
        return _LRN.apply(x, n // 2, float(k), float(alpha), float(beta))
    sq = (x * x).unsqueeze(1)
    pad = n // 2
    sq = F.pad(sq, (0, 0, 0, 0, pad, n - 1 - pad))
    s = F.avg_pool3d(sq, (n, 1, 1), stride=1).squeeze(1) * n
    return x / (k + alpha * s) ** beta


def nchw_f32_to_nhwc(x, dtype, out=None):
    """Contiguous NCHW fp32 image batch (C <= 4) -> the same logical [N, C, H, W] tensor in channels-last memory and ``dtype``
    (fp32 / bf16) with one kernel (native/csrc/layout.hip). Returns None when the kernel does not take the arguments. ``out``: a
    tensor of that shape / dtype / memory format to write into (the static input of a captured training step)."""
    from ..native import lib as L
    if not (L.use_hip(x) and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous() and 1 <= x.shape[1] <= 4
            and dtype in (torch.float32, torch.bfloat16) and x.numel() > 0):
        return None
    n, c, h, w = x.shape
    if out is not None:
        if not (out.shape == x.shape and out.dtype == dtype and out.device == x.device and out.permute(0, 2, 3, 1).is_contiguous()):
            return None
        y = out
    elif c == 1:                     # one channel: both formats are the same memory; strides of a size-1 dim are free
        y = torch.empty((n, h, w, c), device=x.device, dtype=dtype).permute(0, 3, 1, 2)
    else:
        y = torch.empty((n, c, h, w), device=x.device, dtype=dtype, memory_format=torch.channels_last)
    L.check(L.hip().dl4j_nchw_f32_to_nhwc(L.ptr(x), L.ptr(y), L.c_int64(n), L.c_int(c), L.c_int64(h * w), L.c_int(0 if dtype == torch.float32 else 1),
                                          L.stream()), "nchw_f32_to_nhwc")
    return y
