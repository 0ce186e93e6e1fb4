"""Dense (fully connected) layer math on the in-tree MFMA kernels instead of library GEMMs:

    z  = x . Wt^T + b            ops/gemm.py gemm_nt (bias in the epilogue)            Wt = [nOut, nIn], DL4J's 'f'-ordered W in memory
    dx = dz . W_dgrad^T          gemm_nt on the [nIn, nOut] image of W kept by the cast table ("dgrad" replica)
    dW = dz^T . x                the TN form of the LDS-DMA GEMM core when the output fills the chip, else the split-K weight-gradient
                                 kernel (a dense layer is a 1x1 convolution over M "pixels")

Reference: nn/layers/BaseLayer.java:92-111 (backpropGradient: gemm(input, delta, true, false), W.mmul(delta^T)^T) and :322-331
(preOutput), which go to cuBLAS through MmulHelper; here no rocBLAS/hipBLASLt kernel (``Cijk_*``) is launched for these shapes.
"""
from __future__ import annotations

import torch

from . import gemm as G
from . import conv_igemm as IG


def fwd_ok(x2, wt):
    return G.supported(x2, wt) and not (torch.is_grad_enabled() and (x2.requires_grad or wt.requires_grad))


def forward(x2, wt, bias32, act=None):
    return G.gemm_nt(x2, wt, bias=bias32, act=act)


def dx_ok(dz2, w_dgrad):
    return w_dgrad is not None and dz2.dtype == w_dgrad.dtype and G.supported(dz2, w_dgrad)


def dx(dz2, w_dgrad):
    return G.gemm_nt(dz2, w_dgrad)


def dw_ok(dz2, x2, dw_t):
    """dw_t: the [nOut, nIn] fp32 contiguous image of the gradient (= grads['W'].t() for an 'f'-ordered W)."""
    if not (x2.is_cuda and dz2.dtype == x2.dtype and dw_t.dtype == torch.float32 and dw_t.is_contiguous() and x2.is_contiguous()
            and dz2.is_contiguous() and x2.data_ptr() % 16 == 0 and dz2.data_ptr() % 16 == 0):
        return False
    ch = 4 if x2.dtype == torch.float32 else 8
    return x2.dtype in (torch.bfloat16, torch.float16, torch.float32) and x2.shape[1] % ch == 0 and dz2.shape[1] % ch == 0


def _tn_core_wins(nOut, nIn, dtype):
    """Measured (profiles/r3_gemm_layouts.md): the TN kernel on the LDS-DMA core beats the round-1 wgrad kernel when the output alone
    fills the chip (>= 128 tiles of 128 x 128, no split-K: 768 x 3072 51.9 vs 71.2 us, 4096 x 4096 383 vs 527 us); below that the
    round-1 kernel's LDS-reduced split-K atomics are ahead (768 x 768 25 vs 43 us)."""
    return dtype in (torch.bfloat16, torch.float16) and ((nOut + 127) // 128) * ((nIn + 127) // 128) >= 128


def dw_takes_db(dz2, x2):
    """True when dw(..., db_out=...) can accumulate the bias gradient (column sums of dz) in the same launch."""
    nIn, nOut = x2.shape[1], dz2.shape[1]
    return (x2.dtype in (torch.bfloat16, torch.float16)
            and not (_tn_core_wins(nOut, nIn, x2.dtype) and G.supported_ex(dz2.t(), x2.t())))


def dw(dz2, x2, dw_t, prezeroed, db_out=None, db_of_x=False):
    """``db_out`` (fp32, accumulated into): column sums of dz2 - or of x2 with ``db_of_x`` (ops/linear.py passes the gradient in the
    x2 slot to get the [nIn, nOut] image of dW) - from the same launch; only when dw_takes_db()."""
    M, nIn = x2.shape
    nOut = dz2.shape[1]
    if db_out is not None:
        assert dw_takes_db(dz2, x2)
    if _tn_core_wins(nOut, nIn, x2.dtype) and G.supported_ex(dz2.t(), x2.t()):
        # dW[nOut, nIn] = dz^T . x with both operands read K-major (no transposes); accumulates like the atomics of the other path
        G.gemm(dz2.t(), x2.t(), out=dw_t, splits=1, accum=bool(prezeroed))
        return
    IG.wgrad(dz2.view(M, nOut, 1, 1), x2.view(M, nIn, 1, 1), (nOut, nIn, 1, 1), (1, 1), (0, 0), (1, 1), dw_out=dw_t.view(nOut, nIn, 1, 1),
             prezeroed=prezeroed, db_out=db_out, db_of_x=db_of_x)
