"""BatchNorm -> ReLU -> MaxPool as one forward pass and a backward that never materialises the full-resolution activation or its
gradient (native/csrc/bn_pool_fused.hip; bf16, channels-last). The graph fusion planner (nn/graph/computation_graph.py
``_FusedBNPoolNode``) routes the stem of the ImageNet networks here.

Reference: the three layers it replaces - nn/layers/normalization/BatchNormalization.java, nn/layers/ActivationLayer.java,
nn/layers/convolution/subsampling/SubsamplingLayer.java."""
from __future__ import annotations

import ctypes
import os

import torch

from ..native import lib as L

ENABLED = os.environ.get("DL4J_AMD_BN_POOL_FUSED", "1") == "1"


def geometry(shp, kernel, stride, pad):
    """[N, H, W, C, OH, OW, kH, kW, sH, sW, pH, pW] for an NHWC shape tuple (truncate-mode output size)."""
    n, h, w, c = shp
    oh = (h + 2 * pad[0] - kernel[0]) // stride[0] + 1
    ow = (w + 2 * pad[1] - kernel[1]) // stride[1] + 1
    return [int(v) for v in (n, h, w, c, oh, ow, kernel[0], kernel[1], stride[0], stride[1], pad[0], pad[1])]


def _garr(geom):
    return (ctypes.c_int * 12)(*geom)


def usable(x2, geom):
    return (ENABLED and L.use_hip(x2) and x2.dtype == torch.bfloat16 and x2.is_contiguous() and geom[4] > 0 and geom[5] > 0
            and bool(L.hip().dl4j_bn_pool_supported(_garr(geom))))


def forward(x2, scale, shift, geom, train):
    """x2 [N*H*W, C] bf16 (the NHWC tensor flattened), scale / shift [C] fp32 -> (y [N, OH, OW, C] bf16, idx bytes or None)."""
    n, c, oh, ow = geom[0], geom[3], geom[4], geom[5]
    y = torch.empty(n, oh, ow, c, device=x2.device, dtype=x2.dtype)
    idx = torch.empty(n, oh, ow, c, device=x2.device, dtype=torch.uint8) if train else None
    L.check(L.hip().dl4j_bn_relu_pool_fwd(L.ptr(x2), L.ptr(scale.contiguous()), L.ptr(shift.contiguous()), L.ptr(y), L.ptr(idx), _garr(geom),
                                          L.stream()), "bn_relu_pool_fwd")
    return y, idx


def backward(x2, dy, y, idx, gamma, beta, mean, invstd, dgamma_out, dbeta_out, geom):
    """dL/dx2 [N*H*W, C] bf16; dgamma / dbeta written into the given fp32 views (None: skipped)."""
    c = geom[3]
    ws = torch.empty(7 * c, device=x2.device, dtype=torch.float32)
    direct = all(t is None or (t.is_contiguous() and t.dtype == torch.float32) for t in (dgamma_out, dbeta_out))
    dx = torch.empty_like(x2)
    tmp = None if direct else torch.empty(2 * c, device=x2.device, dtype=torch.float32)
    L.check(L.hip().dl4j_bn_pool_bwd(L.ptr(x2), L.ptr(dy), L.ptr(y), L.ptr(idx), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd), L.ptr(ws),
                                     L.ptr(dbeta_out if direct else tmp[:c]), L.ptr(dgamma_out if direct else tmp[c:]), L.ptr(dx), _garr(geom),
                                     L.stream()), "bn_pool_bwd")
    if not direct:
        if dbeta_out is not None:
            dbeta_out.copy_(tmp[:c])
        if dgamma_out is not None:
            dgamma_out.copy_(tmp[c:])
    return dx
