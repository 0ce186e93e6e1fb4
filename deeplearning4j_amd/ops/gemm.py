"""Hand-written MFMA GEMM entry point: y[M, N] = x[M, K] . w[N, K]^T (+ bias), bf16 in / fp32 accumulate / bf16 out.

It is the implicit-GEMM convolution kernel of native/csrc/conv_igemm.hip (32x32x16 bf16 MFMA, LDS double buffering,
XCD-aware tile order, bias epilogue) run as a 1x1 convolution over M "pixels" — the reference's MmulHelper falls back to a naive
one-thread-per-element kernel for bf16 (helpers/cuda_off/MmulHelper.cu:41-106). Plain library GEMMs stay on hipBLASLt
(tools/gemm_microbench.py compares the two); this entry point exists for fused uses and as the hand-written baseline.
"""
from __future__ import annotations

import torch

from . import conv_igemm as IG


def supported(x, w):
    return (x.is_cuda and w.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2
            and x.shape[1] == w.shape[1] and x.shape[1] % 64 == 0 and w.shape[0] % 64 == 0 and x.is_contiguous() and w.is_contiguous()
            and x.numel() < (1 << 31))


def gemm_nt(x, w, bias=None):
    """x [M, K], w [N, K] (row-major, i.e. the transposed-B layout of a dense layer's weight) -> [M, N]."""
    assert supported(x, w), "gemm_nt: bf16 contiguous operands with K and N multiples of 64 are required"
    M, K = x.shape
    N = w.shape[0]
    x4 = x.view(M, 1, 1, K).permute(0, 3, 1, 2)           # logical [M, K, 1, 1], channels-last memory == [M, K]
    w4 = w.view(N, 1, 1, K).permute(0, 3, 1, 2)
    y, _ = IG.forward(x4, w4, bias, (1, 1), (0, 0), (1, 1), want_stats=False)
    return y.permute(0, 2, 3, 1).reshape(M, N)
