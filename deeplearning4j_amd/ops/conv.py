"""conv2d forward / backward dispatch.

Paths (selected per call; analogue of libnd4j's PlatformHelper ``isUsable`` dispatch and the zoo
models' ``cudnnAlgoMode(PREFER_FASTEST)`` — reference: ops/declarable/impl/DeclarableOp.cpp:692-739,
platform/cudnn/conv2d.cu:83):

* ``hip``     – hand-written gfx950 implicit-GEMM MFMA kernels (native/csrc/conv_igemm.hip):
                bf16 NHWC activations, OHWI filters, fp32 accumulation; forward optionally fuses the
                per-channel batch statistics needed by a following BatchNorm into the epilogue.
* ``library`` – MIOpen through ATen (also the CPU / fp64 path used by gradient checks).

Weight layout contract matches DL4J: logical [nOut, nIn, kH, kW] (wFormat 1 of libnd4j conv2d,
generic/nn/convo/conv2d.cpp:46-59).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from ..native import lib as L

_FORCE = os.environ.get("DL4J_AMD_CONV", "auto")  # auto | hip | library


def _hip_eligible(x, w, stride, padding, dilation):
    if _FORCE == "library" or not L.use_hip(x, w):
        return False
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        return False
    try:
        from . import conv_igemm
    except Exception:
        return False
    return conv_igemm.supported(x, w, stride, padding, dilation)


def conv2d_forward(x, w, b, stride, padding, dilation, stats_out=None):
    """x: [N,C,H,W] logical (channels_last memory on GPU). Returns z (same memory format)."""
    if _hip_eligible(x, w, stride, padding, dilation):
        from . import conv_igemm
        return conv_igemm.forward(x, w, b, stride, padding, dilation, stats_out)
    return F.conv2d(x, w, b, tuple(stride), tuple(padding), tuple(dilation))


def conv2d_backward(dz, x, w, stride, padding, dilation, need_dx=True, need_dw=True, need_db=False):
    """Returns (dx, dW, db); dW has logical shape [nOut, nIn, kH, kW]."""
    if _hip_eligible(x, w, stride, padding, dilation):
        from . import conv_igemm
        return conv_igemm.backward(dz, x, w, stride, padding, dilation, need_dx, need_dw, need_db)
    if x.is_cuda and not dz.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last):
        dz = dz.contiguous(memory_format=torch.channels_last)
    dx, dw, db = torch.ops.aten.convolution_backward(
        dz, x, w, [w.shape[0]] if need_db else None, list(stride), list(padding), list(dilation), False, [0, 0], 1,
        [bool(need_dx), bool(need_dw), bool(need_db)])
    return dx, dw, db
