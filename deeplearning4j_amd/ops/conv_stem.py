"""Launchers of the 3-channel stem convolution kernels (native/csrc/conv_stem.hip): 7x7 stride-2 conv with 64 filters on
bf16 NHWC input (the zoo ResNet-50 stem, zoo/model/ResNet50.java), forward with fused BatchNorm statistics, and weight
gradient. Shapes are validated here before every launch."""
from __future__ import annotations

import torch

from ..native import lib as L
from .conv_igemm import StatPartials

FWD_ROWS_PER_BLOCK = None       # None: sized so that ~512 workgroups cover the batch (measured best, tools/stem_microbench.py)
WGRAD_ROWS_PER_BLOCK = None
_SCRATCH = {}


def supported(x, w, stride, padding, dilation):
    if not (L.use_hip(x, w) and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 4):
        return False
    if x.shape[1] != 3 or tuple(w.shape) != (64, 3, 7, 7) or tuple(stride) != (2, 2) or tuple(dilation) != (1, 1):
        return False
    if padding[0] != padding[1] or not 0 <= padding[0] <= 7 or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    ow = (x.shape[3] + 2 * padding[1] - 7) // 2 + 1
    oh = (x.shape[2] + 2 * padding[0] - 7) // 2 + 1
    return 0 < ow <= 128 and oh > 0 and x.numel() < (1 << 31)


def _out_hw(x, padding):
    return (x.shape[2] + 2 * padding[0] - 7) // 2 + 1, (x.shape[3] + 2 * padding[1] - 7) // 2 + 1


def _rows_per_block(forced, N, OH):
    if forced:
        return int(forced)
    nrb = max(1, min(OH, round(512 / N)))          # workgroups per image: each walks down its rows with an LDS ring
    return -(-OH // nrb)


def pack_weights(w):
    """[64, 3, 7, 7] (any memory format) -> bf16 [64, 7, 8, 4]: taps padded 7 -> 8, channels 3 -> 4 (zeros)."""
    wp = torch.zeros(64, 7, 8, 4, device=w.device, dtype=torch.bfloat16)
    wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    return wp


def forward(x, w, bias, stride, padding, dilation, want_stats=False):
    assert supported(x, w, stride, padding, dilation)
    N, _, H, W = x.shape
    OH, OW = _out_hw(x, padding)
    y = torch.empty((N, 64, OH, OW), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
    rpb = _rows_per_block(FWD_ROWS_PER_BLOCK, N, OH)
    nblk = N * ((OH + rpb - 1) // rpb)
    stats = torch.empty(nblk * 128, device=x.device, dtype=torch.float32) if want_stats else None
    b = None if bias is None else bias.reshape(-1).float().contiguous()
    assert b is None or b.numel() == 64
    wp = pack_weights(w)
    L.check(L.hip().dl4j_conv_c3_fwd(L.ptr(x), L.ptr(wp), L.ptr(b), L.ptr(y), L.ptr(stats), L.c_int(N), L.c_int(H), L.c_int(W),
                                     L.c_int(OH), L.c_int(OW), L.c_int(7), L.c_int(7), L.c_int(2), L.c_int(int(padding[0])), L.c_int(rpb),
                                     L.stream()), "conv_c3_fwd")
    return y, (StatPartials(stats, nblk, 64) if want_stats else None)


def wgrad(dz, x, stride, padding, dilation, dw_out=None):
    """dW of the stem conv as fp32 [64, 3, 7, 7]; written into ``dw_out`` (contiguous fp32 view) when given -> returns None."""
    N, _, H, W = x.shape
    OH, OW = _out_hw(x, padding)
    assert tuple(dz.shape) == (N, 64, OH, OW) and dz.dtype == torch.bfloat16
    dzc = dz if dz.is_contiguous(memory_format=torch.channels_last) else dz.contiguous(memory_format=torch.channels_last)
    key = (x.device.index, torch.cuda.current_stream().cuda_stream)
    scratch = _SCRATCH.get(key)
    if scratch is None:
        scratch = _SCRATCH[key] = torch.zeros(64 * 7 * 32, device=x.device, dtype=torch.float32)   # the kernel leaves it zeroed
    tgt = dw_out if dw_out is not None else torch.empty(64, 3, 7, 7, device=x.device, dtype=torch.float32)
    assert tgt.dtype == torch.float32 and tgt.is_contiguous() and tgt.numel() == 64 * 147
    L.check(L.hip().dl4j_conv_c3_wgrad(L.ptr(dzc), L.ptr(x), L.ptr(scratch), L.ptr(tgt), L.c_int(N), L.c_int(H), L.c_int(W), L.c_int(OH),
                                       L.c_int(OW), L.c_int(7), L.c_int(7), L.c_int(2), L.c_int(int(padding[0])),
                                       L.c_int(_rows_per_block(WGRAD_ROWS_PER_BLOCK, N, OH)), L.c_int(0), L.stream()), "conv_c3_wgrad")
    return None if dw_out is not None else tgt


def wgrad_bnpool(lazy, x, stride, padding, dilation, dw_out=None):
    """dW of the stem conv when its output gradient is a bn_pool.PoolBnGrad: the gradient rows are recomputed inside the kernel's loader
    from the pooled gradient, the argmax bytes and the convolution's own output (never written to memory)."""
    N, _, H, W = x.shape
    OH, OW = _out_hw(x, padding)
    g = lazy.geom
    assert lazy.fusable_3x3s2() and (g[0], g[1], g[2]) == (N, OH, OW) and lazy.x2.dtype == torch.bfloat16
    key = (x.device.index, torch.cuda.current_stream().cuda_stream)
    scratch = _SCRATCH.get(key)
    if scratch is None:
        scratch = _SCRATCH[key] = torch.zeros(64 * 7 * 32, device=x.device, dtype=torch.float32)
    tgt = dw_out if dw_out is not None else torch.empty(64, 3, 7, 7, device=x.device, dtype=torch.float32)
    assert tgt.dtype == torch.float32 and tgt.is_contiguous() and tgt.numel() == 64 * 147
    coef = lazy.ws[3 * 64:6 * 64]
    L.check(L.hip().dl4j_conv_c3_wgrad_bnpool(L.ptr(lazy.x2), L.ptr(lazy.dy), L.ptr(lazy.idx), L.ptr(coef), L.c_int(g[4]), L.c_int(g[5]),
                                              L.c_int(g[10]), L.ptr(x), L.ptr(scratch), L.ptr(tgt), L.c_int(N), L.c_int(H), L.c_int(W),
                                              L.c_int(OH), L.c_int(OW), L.c_int(7), L.c_int(7), L.c_int(2), L.c_int(int(padding[0])),
                                              L.c_int(_rows_per_block(WGRAD_ROWS_PER_BLOCK, N, OH)), L.c_int(0), L.stream()),
            "conv_c3_wgrad_bnpool")
    return None if dw_out is not None else tgt
