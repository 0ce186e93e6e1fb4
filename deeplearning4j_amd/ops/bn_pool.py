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
# hand the input gradient on as a PoolBnGrad (see below). Measured on the ResNet-50 step: 9.38 ms with, 9.28 ms without - the loader's
# gather (36 loads and ~1000 VALU instructions per thread and row at one wave per SIMD) costs more than the 411 MB round trip it
# saves, so the separate element pass stays the default; the fused form is tested (tests/test_bn_pool_fused_gpu.py) and selectable.
LAZY_DX = os.environ.get("DL4J_AMD_BN_POOL_LAZY_DX", "0") == "1"


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


class PoolBnGrad:
    """The input gradient of the fused node, NOT yet computed: the reductions (pass 1) and the per-channel coefficients are done,
    the per-element pass is left to the consumer. The stem convolution's weight-gradient kernel recomputes the gradient rows in its
    own loader (ops/conv_stem.wgrad_bnpool -> dl4j_conv_c3_wgrad_bnpool), so the 411 MB tensor of the ImageNet stem is neither
    written nor read back; anybody else calls ``materialize()`` (pass 2 as its own launch)."""
    __slots__ = ("x2", "dy", "idx", "ws", "geom", "shp", "_from2d", "_dx")

    def __init__(self, x2, dy, idx, ws, geom, shp, from2d):
        self.x2, self.dy, self.idx, self.ws, self.geom, self.shp, self._from2d, self._dx = x2, dy, idx, ws, geom, shp, from2d, None

    @property
    def shape(self):
        n, h, w, c = self.geom[0], self.geom[1], self.geom[2], self.geom[3]
        return torch.Size((n, c, h, w))

    @property
    def dtype(self):
        return self.x2.dtype

    def fusable_3x3s2(self):
        g = self.geom
        return g[3] == 64 and tuple(g[6:10]) == (3, 3, 2, 2) and g[10] == g[11] and 0 <= g[10] <= 2

    def materialize(self):
        if self._dx is None:
            dx2 = torch.empty_like(self.x2)
            L.check(L.hip().dl4j_bn_pool_bwd_dx(L.ptr(self.x2), L.ptr(self.dy), L.ptr(self.idx), L.ptr(self.ws), L.ptr(dx2), _garr(self.geom),
                                                L.stream()), "bn_pool_bwd_dx")
            self._dx = self._from2d(dx2, self.shp)
        return self._dx

    def __add__(self, other):
        o = other.materialize() if isinstance(other, PoolBnGrad) else other
        return self.materialize() + o

    __radd__ = __add__


def backward_lazy(x2, dy, y, idx, gamma, beta, mean, invstd, dgamma_out, dbeta_out, geom, shp, from2d):
    """Passes 1 + coefficients now (dgamma / dbeta are final), the element pass later: returns a PoolBnGrad."""
    c = geom[3]
    ws = torch.empty(7 * c, device=x2.device, dtype=torch.float32)
    direct = all(t is None or (t.is_contiguous() and t.dtype == torch.float32) for t in (dgamma_out, dbeta_out))
    tmp = None if direct else torch.empty(2 * c, device=x2.device, dtype=torch.float32)
    L.check(L.hip().dl4j_bn_pool_bwd_reduce(L.ptr(dy), L.ptr(y), L.ptr(idx), L.ptr(gamma), L.ptr(beta), L.ptr(mean), L.ptr(invstd), L.ptr(ws),
                                            L.ptr(dbeta_out if direct else tmp[:c]), L.ptr(dgamma_out if direct else tmp[c:]), _garr(geom),
                                            L.stream()), "bn_pool_bwd_reduce")
    if not direct:
        if dbeta_out is not None:
            dbeta_out.copy_(tmp[:c])
        if dgamma_out is not None:
            dgamma_out.copy_(tmp[c:])
    return PoolBnGrad(x2, dy, idx, ws, geom, shp, from2d)
