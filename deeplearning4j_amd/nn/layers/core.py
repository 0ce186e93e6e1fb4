"""Hand-written (non-autograd) runtime layers: dense/output/loss, activation, dropout, embedding,
batch-norm, convolution, pooling, padding, global pooling.

Reference behaviours: nn/layers/BaseLayer.java:92-111,322-331 (dense fwd/bwd),
nn/layers/convolution/ConvolutionLayer.java (im2col+gemm there; one fused conv op here),
nn/layers/normalization/BatchNormalization.java, nn/layers/convolution/subsampling/SubsamplingLayer.java,
nn/layers/BaseOutputLayer.java, nn/layers/pooling/GlobalPoolingLayer.java.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from ...nd4j.activations import ActivationIdentity, ActivationReLU, ActivationSoftmax
from ...ops import batchnorm as BN
from ...ops import pooling as POOL
from ...ops import conv as CONV
from ...ops import dense as DENSE
from ..conf import layers as C
from ..conf.base import ConvolutionMode, PoolingType, InputType
from .base import LayerImpl, AutogradLayerImpl


# ------------------------------------------------------------------ dense family
class DenseLayerImpl(LayerImpl):
    def compute_replicas(self):
        # "dgrad": the [nIn, nOut] row-major image of W for the input-gradient GEMM (ops/dense.py)
        return [("W", "plain"), ("W", "dgrad")] + ([("b", "plain")] if "b" in self.params else [])

    def _wt(self):
        # flat layout is 'f' order [nIn, nOut]  ==  contiguous [nOut, nIn]; use it as linear weight
        return self.w("W").t()

    def _bias32(self):
        if "b" not in self.params:
            return None
        b = self.params["b"]
        return (b if b.dtype == torch.float32 else self.w("b").float()).reshape(-1)

    def pre_output(self, x):
        W = self._wt()  # [nOut, nIn] contiguous
        x2 = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
        if x.is_cuda and DENSE.fwd_ok(x2, W):
            z = DENSE.forward(x2, W, self._bias32())
            if x.dim() != 2:
                z = z.view(*x.shape[:-1], W.shape[0])
        else:
            b = self.w("b").reshape(-1) if "b" in self.params else None
            z = F.linear(x, W, b)
        if getattr(self.conf, "hasLayerNorm", False):
            self._ln_in = z
            z = F.layer_norm(z, (z.shape[-1],), self.w("g").reshape(-1), None, 1e-5)
        return z

    def forward(self, x, train=False):
        x = self._dropout_in(x, train)
        if x.dtype != self.ctx.cdtype and x.is_floating_point():
            x = x.to(self.ctx.cdtype)
        self.input = x
        z = self.pre_output(x)
        self.z = z
        return self.conf.activationFn.forward(z, train)

    def _backward_from_delta(self, dz, need_dx=True):
        x = self.input
        if getattr(self.conf, "hasLayerNorm", False):
            zi = self._ln_in.detach().requires_grad_(True)
            g = self.w("g").reshape(-1).detach().requires_grad_(True)
            with torch.enable_grad():
                zz = F.layer_norm(zi, (zi.shape[-1],), g, None, 1e-5)
            dz, dg = torch.autograd.grad(zz, [zi, g], dz)
            self.grads["g"].copy_(dg.reshape(self.grads["g"].shape))
        x2 = x.reshape(-1, x.shape[-1])
        d2 = dz.reshape(-1, dz.shape[-1])
        # dW in 'f' layout == [nOut, nIn] contiguous = dz^T x  (BaseLayer.java:96 gemm(input, delta, true, false))
        if d2.dtype != x2.dtype:
            d2 = d2.to(x2.dtype)
        gw = self.grads["W"].t()
        if x2.is_cuda and DENSE.dw_ok(d2, x2, gw):
            DENSE.dw(d2, x2, gw, getattr(self.ctx, "grads_zeroed", False))
        else:
            gw.copy_(torch.matmul(d2.t(), x2))
        if "b" in self.grads:
            self.grads["b"].copy_(d2.sum(0, dtype=torch.float32 if d2.dtype in (torch.bfloat16, torch.float16) else None).reshape(1, -1))
        if not need_dx:
            return None
        wd = self.cparams.get("W_dgrad")
        if x2.is_cuda and DENSE.dx_ok(d2, wd):
            dx = DENSE.dx(d2, wd).view(*dz.shape[:-1], wd.shape[0])
        else:
            dx = torch.matmul(dz, self._wt())
        return self._dropout_back(dx)

    def backward(self, eps):
        dz = self.conf.activationFn.backprop(self.z, eps)
        return self._backward_from_delta(dz, getattr(self, "needs_input_grad", True))

    def clear(self):
        super().clear()
        self.z = None


class OutputLayerImpl(DenseLayerImpl):
    """Dense + loss. ``compute_score_and_delta`` is the fused softmax+MCXENT path when applicable."""
    is_output = True

    def setLabels(self, labels, mask=None):
        self.labels, self.labelMask = labels, mask

    def _labels2d(self):
        return self.labels, self.labelMask

    def _z2d(self):
        return self.z

    def compute_score_sum(self):
        """Sum of per-example scores (no regularisation, not divided by minibatch)."""
        labels, mask = self._labels2d()
        s, dz = self.conf.lossFn.computeGradientAndScore(labels, self._z2d(), self.conf.activationFn, mask, False)
        self._delta = dz
        return s

    def score_array(self):
        labels, mask = self._labels2d()
        return self.conf.lossFn.computeScoreArray(labels, self._z2d(), self.conf.activationFn, mask)

    def backward(self, eps=None):
        dz = self._delta
        self._delta = None
        return self._backward_from_delta(self._delta_to_input_shape(dz), getattr(self, "needs_input_grad", True))

    def _delta_to_input_shape(self, dz):
        return dz


class RnnOutputLayerImpl(OutputLayerImpl):
    """Output layer applied per time step; activations [mb, nOut, T] (RnnOutputLayer.java)."""

    def forward(self, x, train=False):
        # x: [mb, nIn, T] -> 2d [mb*T, nIn]
        self._mb, self._T = x.shape[0], x.shape[2]
        x2 = x.permute(0, 2, 1).reshape(-1, x.shape[1])
        y2 = super().forward(x2, train)
        return y2.reshape(self._mb, self._T, -1).permute(0, 2, 1)

    def _labels2d(self):
        l = self.labels.permute(0, 2, 1).reshape(-1, self.labels.shape[1])
        m = self.labelMask
        if m is not None:
            m = m.reshape(-1, 1)
        return l, m

    def backward(self, eps=None):
        dx2 = super().backward(eps)
        if dx2 is None:
            return None
        return dx2.reshape(self._mb, self._T, -1).permute(0, 2, 1)


class LossLayerImpl(LayerImpl):
    is_output = True

    def forward(self, x, train=False):
        self.input = x
        self.z = x
        return self.conf.activationFn.forward(x, train) if getattr(self.conf, "activationFn", None) else x

    def setLabels(self, labels, mask=None):
        self.labels, self.labelMask = labels, mask

    def _flatten(self, t):
        if t is None:
            return None
        if t.dim() == 3:
            return t.permute(0, 2, 1).reshape(-1, t.shape[1])
        if t.dim() == 4:
            return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
        if t.dim() == 5:
            return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1])
        return t

    def compute_score_sum(self):
        z2, l2 = self._flatten(self.z), self._flatten(self.labels)
        m = self.labelMask
        if m is not None and m.dim() > 1 and self.z.dim() > 2:
            m = m.reshape(-1, 1) if m.numel() == z2.shape[0] else self._flatten(m)
        act = self.conf.activationFn or ActivationIdentity()
        s, dz = self.conf.lossFn.computeGradientAndScore(l2, z2, act, m, False)
        self._delta = dz
        return s

    def score_array(self):
        act = self.conf.activationFn or ActivationIdentity()
        return self.conf.lossFn.computeScoreArray(self._flatten(self.labels), self._flatten(self.z), act, None)

    def backward(self, eps=None):
        dz = self._delta
        z = self.z
        if z.dim() == 3:
            return dz.reshape(z.shape[0], z.shape[2], z.shape[1]).permute(0, 2, 1)
        if z.dim() == 4:
            return dz.reshape(z.shape[0], z.shape[2], z.shape[3], z.shape[1]).permute(0, 3, 1, 2)
        if z.dim() == 5:
            return dz.reshape(z.shape[0], z.shape[2], z.shape[3], z.shape[4], z.shape[1]).permute(0, 4, 1, 2, 3)
        return dz


class ActivationLayerImpl(LayerImpl):
    def forward(self, x, train=False):
        x = self._dropout_in(x, train)
        self.input = x
        return self.conf.activationFn.forward(x, train)

    def backward(self, eps):
        return self._dropout_back(self.conf.activationFn.backprop(self.input, eps))


class DropoutLayerImpl(LayerImpl):
    def forward(self, x, train=False):
        return self._dropout_in(x, train)

    def backward(self, eps):
        return self._dropout_back(eps)


class EmbeddingLayerImpl(LayerImpl):
    def compute_replicas(self):
        return [("W", "plain")]

    def forward(self, x, train=False):
        idx = x.reshape(x.shape[0], -1)[:, 0].long() if x.dim() > 1 else x.long()
        self._idx = idx
        W = self.w("W")  # logical [nIn, nOut], 'f' strides
        z = W[idx]
        if "b" in self.params:
            z = z + self.w("b").reshape(1, -1)
        self.z = z
        return self.conf.activationFn.forward(z, train)

    def backward(self, eps):
        dz = self.conf.activationFn.backprop(self.z, eps)
        g = self.grads["W"]
        g.zero_()
        g.index_add_(0, self._idx, dz.to(g.dtype))
        if "b" in self.grads:
            self.grads["b"].copy_(dz.sum(0).reshape(1, -1))
        return None


class EmbeddingSequenceLayerImpl(LayerImpl):
    def compute_replicas(self):
        return [("W", "plain")]

    def forward(self, x, train=False):
        # x: [mb, T] or [mb, 1, T] integer indices -> [mb, nOut, T]
        if x.dim() == 3:
            x = x[:, 0, :]
        idx = x.long()
        self._idx = idx
        z = self.w("W")[idx]  # [mb, T, nOut]
        if "b" in self.params:
            z = z + self.w("b").reshape(1, 1, -1)
        z = z.permute(0, 2, 1)
        self.z = z
        return self.conf.activationFn.forward(z, train)

    def backward(self, eps):
        dz = self.conf.activationFn.backprop(self.z, eps).permute(0, 2, 1)  # [mb, T, nOut]
        g = self.grads["W"]
        g.zero_()
        g.index_add_(0, self._idx.reshape(-1), dz.reshape(-1, dz.shape[-1]).to(g.dtype))
        if "b" in self.grads:
            self.grads["b"].copy_(dz.sum((0, 1)).reshape(1, -1))
        return None


# ------------------------------------------------------------------ batch norm
class BatchNormalizationImpl(LayerImpl):
    """Fused BN (+ optional residual add and ReLU when driven by the graph fusion planner)."""

    def __init__(self, conf, ctx):
        super().__init__(conf, ctx)
        self.conv_bias = None      # set by the fusion planner when the producing conv's bias is folded in
        self._ctx = None

    def _views(self):
        p, g = self.params, self.grads
        vkey = "log10stdev" if self.conf.useLogStd else "var"
        gamma = p["gamma"].reshape(-1) if "gamma" in p else None
        beta = p["beta"].reshape(-1) if "beta" in p else None
        return gamma, beta, p["mean"].reshape(-1), p[vkey].reshape(-1), \
            (g["mean"].reshape(-1) if "mean" in g else None), (g[vkey].reshape(-1) if vkey in g else None)

    @staticmethod
    def _to2d(x):
        if x.dim() == 2:
            return x, None
        if x.dim() == 4:
            N, Cc, H, W = x.shape
            return x.permute(0, 2, 3, 1).reshape(-1, Cc), (N, H, W, Cc)
        if x.dim() == 3:   # [mb, C, T]
            N, Cc, T = x.shape
            return x.permute(0, 2, 1).reshape(-1, Cc), (N, T, Cc)
        if x.dim() == 5:
            N, Cc, D, H, W = x.shape
            return x.permute(0, 2, 3, 4, 1).reshape(-1, Cc), (N, D, H, W, Cc)
        raise ValueError("BatchNormalization: unsupported rank %d" % x.dim())

    @staticmethod
    def _from2d(y2, shp):
        if shp is None:
            return y2
        if len(shp) == 4:
            return y2.view(shp).permute(0, 3, 1, 2)
        if len(shp) == 3:
            return y2.view(shp).permute(0, 2, 1)
        return y2.view(shp).permute(0, 4, 1, 2, 3)

    def finalize_request(self):
        """Everything the producing conv needs to run this layer's training-mode finalize inside its own launch."""
        gamma, beta, gmean, gvar, dgm, dgv = self._views()
        cb = self.conv_bias() if self.conv_bias is not None else None
        return BN.FinalizeRequest(gamma, beta, cb, gmean, gvar, dgm, dgv, self.conf.eps, self.conf.decay, self.conf.useLogStd)

    def coefficients(self, x2, train):
        """(scale, shift) for this call; in training also stores mean/invstd and writes stat gradients."""
        gamma, beta, gmean, gvar, dgm, dgv = self._views()
        cb = self.conv_bias() if self.conv_bias is not None else None
        if train:
            pre = self.producer.out_sums if getattr(self, "producer", None) is not None else None
            mean, invstd, scale, shift = BN.batch_stats(x2, gamma, beta, self.conf.eps, self.conf.decay, gmean, gvar,
                                                        self.conf.useLogStd, cb, dgm, dgv, sums=pre)
            self._mean, self._invstd = mean, invstd
            return scale, shift
        return BN.infer_coeffs(gamma, beta, gmean, gvar, self.conf.eps, self.conf.useLogStd, cb)

    def forward(self, x, train=False, relu=False, residual=None, res_coeffs=None):
        x2, shp = self._to2d(x)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        scale, shift = self.coefficients(x2, train)
        r2 = None
        if residual is not None:
            r2, _ = self._to2d(residual)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        # the reference layer never applies its configured activation (BatchNormalization.java:379 activate() == preOutput);
        # ReLU only comes from the graph fusion planner
        fuse_relu = relu
        rs, rt = res_coeffs if res_coeffs is not None else (None, None)
        bits = None
        if train and fuse_relu and residual is not None and BN.mask_supported(x2):
            y2, bits = BN.apply(x2, scale, shift, r2, rs, rt, fuse_relu, want_mask=True)
        else:
            y2 = BN.apply(x2, scale, shift, r2, rs, rt, fuse_relu)
        if train:
            self._x2, self._shp = x2, shp
            # without a residual the ReLU mask is recomputable from x (x*scale+shift > 0): the backward kernels then skip
            # re-reading the output tensor; with a residual the mask is kept as bits (1/16 of the bf16 output) or, on the
            # library / CPU path, read back from the stored output
            self._mask_bits = bits
            self._out2 = y2 if (fuse_relu and residual is not None and bits is None) else None
            self._mask_coeffs = (scale, shift) if (fuse_relu and residual is None) else None
            self._want_dz = residual is not None
        self._z2 = None
        return self._from2d(y2, shp)

    def backward_sink(self):
        """What the data-gradient launch producing this layer's dy needs to emit the backward partial sums in its epilogue
        (ops/conv_igemm.BnBackwardSink), or None when this call's state does not allow it (CPU / fp32 / stored-output mask)."""
        x2 = getattr(self, "_x2", None)
        if x2 is None or not x2.is_cuda or x2.dtype != torch.bfloat16 or self._out2 is not None or self._shp is None \
                or len(self._shp) != 4 or getattr(self, "_ext_mask", False):
            return None
        from ...ops.conv_igemm import BnBackwardSink
        mc = getattr(self, "_mask_coeffs", None)
        dg = self.grads["gamma"].reshape(-1) if "gamma" in self.grads else None
        db = self.grads["beta"].reshape(-1) if "beta" in self.grads else None
        return BnBackwardSink(x2, self._mean, self._invstd, mc[0] if mc else None, mc[1] if mc else None,
                              getattr(self, "_mask_bits", None), dbeta_out=db, dgamma_out=dg)

    def backward(self, eps, return_dz=False, lazy_dz=False, ext_mask_bits=None):
        """``lazy_dz``: return dz = eps * [out > 0] as a MaskedGrad (eps + the saved mask bits) instead of writing it.
        ``ext_mask_bits``: eps is to be read through these mask bits (the shortcut BatchNorm of a downsampling block receives
        the block's masked gradient without it ever being materialised)."""
        from ...ops.conv_igemm import MaskedGrad
        d2, _ = self._to2d(eps)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        if d2.dtype != self._x2.dtype:
            d2 = d2.to(self._x2.dtype)
        gamma = self.params["gamma"].reshape(-1) if "gamma" in self.params else None
        dg = self.grads["gamma"].reshape(-1) if "gamma" in self.grads else None
        db = self.grads["beta"].reshape(-1) if "beta" in self.grads else None
        bits = getattr(self, "_mask_bits", None)
        if ext_mask_bits is not None:
            assert bits is None and self._out2 is None and getattr(self, "_mask_coeffs", None) is None
            bits = ext_mask_bits
        want_dz = return_dz or self._want_dz
        lazy = lazy_dz and want_dz and bits is not None and MaskedGrad.ok(eps, bits)
        sink, self._bwd_sink = getattr(self, "_bwd_sink", None), None
        pre = (sink.partial, sink.gy, sink.sums) if (sink is not None and sink.partial is not None) else None
        dx2, dz2 = BN.backward(self._x2, d2, self._out2, self._mean, self._invstd, gamma, dg, db,
                               want_dz=want_dz and not lazy, mask_coeffs=getattr(self, "_mask_coeffs", None),
                               mask_bits=bits, pre_partials=pre)
        dx = self._from2d(dx2, self._shp)
        if lazy:
            return dx, MaskedGrad(eps, bits)
        if want_dz:
            return dx, self._from2d(dz2, self._shp)
        return dx

    def clear(self):
        super().clear()
        self._x2 = self._out2 = self._z2 = self._mask_coeffs = self._mask_bits = self._bwd_sink = None


# ------------------------------------------------------------------ convolution
_STAT_PIVOT = __import__("os").environ.get("DL4J_AMD_BN_STAT_PIVOT", "1") == "1"


class ConvolutionLayerImpl(LayerImpl):
    def __init__(self, conf, ctx):
        super().__init__(conf, ctx)
        self.fold_bias = False   # bias folded into a following BatchNorm (fusion planner)
        self.want_stats = False  # emit per-channel sum/sumsq for the following BatchNorm (conv epilogue fusion)
        self.out_sums = None
        self.extra_pad = (0, 0)  # symmetric zero padding of a preceding ZeroPaddingLayer absorbed by the fusion planner

    def compute_replicas(self):
        r = [("W", "ohwi"), ("W", "dgrad")]
        if "b" in self.params:
            r.append(("b", "plain"))
        return r

    def _padding(self, x):
        c = self.conf
        if (c.convolutionMode or ConvolutionMode.Truncate) == ConvolutionMode.Same:
            ph = C.same_pad(x.shape[2], c.kernelSize[0], c.stride[0], c.dilation[0])
            pw = C.same_pad(x.shape[3], c.kernelSize[1], c.stride[1], c.dilation[1])
            if ph[0] == ph[1] and pw[0] == pw[1]:
                return (ph[0], pw[0]), None
            return (0, 0), (pw[0], pw[1], ph[0], ph[1])
        return (c.padding[0] + self.extra_pad[0], c.padding[1] + self.extra_pad[1]), None

    def forward(self, x, train=False):
        x = self._dropout_in(x, train)
        if x.dtype != self.ctx.cdtype:
            x = x.to(self.ctx.cdtype)
        pad, extra = self._padding(x)
        if extra is not None:
            x = F.pad(x, extra)
        self._extra = extra
        if self.ctx.channels_last and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        self.input, self._pad = x, pad
        c = self.conf
        b = None if (self.fold_bias or "b" not in self.params) else self.w("b").reshape(-1)
        fin = pivot = None
        if self.want_stats and train and getattr(self, "stats_consumer", None) is not None:
            fin = self.stats_consumer.finalize_request()
            # statistics pivot = the consumer's running mean (minus the folded bias): sums of (z - pivot) do not cancel
            if _STAT_PIVOT and fin.gmean is not None and fin.ok():
                pivot = (fin.gmean, fin.conv_bias)
        z, self.out_sums = CONV.conv2d_forward(x, self.w("W"), b, c.stride, pad, c.dilation,
                                               want_stats=self.want_stats and train, bn_fin=fin, stat_pivot=pivot)
        act = c.activationFn
        if act is None or isinstance(act, ActivationIdentity):
            self.z = None
            return z
        self.z = z
        return act.forward(z, train)

    def supports_accum(self):
        """True when backward(eps, accum=g) can add the input gradient into ``g`` (graph fan-out accumulation)."""
        return self._extra is None and self.conf.idropout is None

    def backward(self, eps, accum=None, bn_sink=None):
        """``accum``: gradient already collected for this layer's input — a tensor (added in place by the dgrad epilogue) or a
        MaskedGrad (added from its source through the mask bits). ``bn_sink``: see ops/conv_igemm.BnBackwardSink."""
        from ...ops.conv_igemm import MaskedGrad
        from ...ops.bn_pool import PoolBnGrad
        c = self.conf
        if isinstance(eps, PoolBnGrad):
            # the gradient of a fused BatchNorm -> ReLU -> MaxPool node, not evaluated yet: the stem weight-gradient kernel recomputes it
            # in its loader when nothing else needs it (first layer: no input gradient, identity activation, no bias gradient)
            if (self.z is None and accum is None and not getattr(self, "needs_input_grad", True)
                    and not ("b" in self.params and not self.fold_bias) and eps.fusable_3x3s2()
                    and CONV.stem_wgrad_bnpool(eps, self.input, self.w("W"), c.stride, self._pad, c.dilation, self.grads["W"])):
                return None
            eps = eps.materialize()
        dz = eps if self.z is None else c.activationFn.backprop(self.z, eps)
        if dz.dtype != self.input.dtype:
            dz = dz.to(self.input.dtype)
        has_b = "b" in self.params and not self.fold_bias
        need_dx = getattr(self, "needs_input_grad", True)
        masked = accum if isinstance(accum, MaskedGrad) else None
        if masked is not None:
            accum = None
        dx, dW, db = CONV.conv2d_backward(dz, self.input, self.w("W"), c.stride, self._pad, c.dilation,
                                          need_dx, True, has_b, w_dgrad=self.cparams.get("W_dgrad"),
                                          dw_out=self.grads["W"], dx_accum=accum if need_dx else None,
                                          dw_prezeroed=getattr(self.ctx, "grads_zeroed", False),
                                          dx_acc_masked=masked if need_dx else None, bn_sink=bn_sink if need_dx else None)
        if dW is not None:
            self.grads["W"].copy_(dW)
        if "b" in self.grads:
            if has_b:
                self.grads["b"].copy_(db.reshape(1, -1))
            # else: analytically zero (d(BN)/d(bias) == 0 when the bias is folded into the following
            # BatchNorm); the flat gradient vector is zero-initialised and this slot is never written
        if dx is None:
            return None
        if self._extra is not None:
            l, r, t, bm = self._extra
            dx = dx[:, :, t:dx.shape[2] - bm, l:dx.shape[3] - r]
        return self._dropout_back(dx)

    def clear(self):
        super().clear()
        self.z = None


class SubsamplingLayerImpl(LayerImpl):
    def forward(self, x, train=False):
        x = self._dropout_in(x, train)
        c = self.conf
        extra = None
        if (c.convolutionMode or ConvolutionMode.Truncate) == ConvolutionMode.Same:
            ph = C.same_pad(x.shape[2], c.kernelSize[0], c.stride[0], c.dilation[0])
            pw = C.same_pad(x.shape[3], c.kernelSize[1], c.stride[1], c.dilation[1])
            if ph[0] == ph[1] and pw[0] == pw[1]:
                pad = (ph[0], pw[0])
            else:
                pad = (0, 0)
                extra = (pw[0], pw[1], ph[0], ph[1])
                fill = float("-inf") if c.poolingType == PoolingType.MAX else 0.0
                x = F.pad(x, extra, value=fill)
        else:
            pad = tuple(c.padding)
        self._extra = extra
        mode = {"MAX": "max", "AVG": "avg", "PNORM": "pnorm", "SUM": "avg"}[c.poolingType]
        y, self._pctx = POOL.pool2d_forward(x, mode, c.kernelSize, c.stride, pad, c.dilation,
                                            avg_exclude_pad=not c.avgPoolIncludePadInDivisor, pnorm=c.pnorm or 2,
                                            train=train)
        if c.poolingType == "SUM":
            y = y * (c.kernelSize[0] * c.kernelSize[1])
        return y

    def backward(self, eps):
        c = self.conf
        if c.poolingType == "SUM":
            eps = eps * (c.kernelSize[0] * c.kernelSize[1])
        dx = POOL.pool2d_backward(eps, self._pctx)
        if self._extra is not None:
            l, r, t, bm = self._extra
            dx = dx[:, :, t:dx.shape[2] - bm, l:dx.shape[3] - r]
        return self._dropout_back(dx)

    def clear(self):
        super().clear()
        self._pctx = None


class ZeroPaddingLayerImpl(LayerImpl):
    passthrough = False          # set by the graph fusion planner when the consuming convolution pads inside its own kernel

    def forward(self, x, train=False):
        if self.passthrough:
            return x
        t, b, l, r = self.conf.padding
        y = F.pad(x, (l, r, t, b))
        if self.ctx.channels_last and x.is_cuda:
            y = y.contiguous(memory_format=torch.channels_last)
        return y

    def backward(self, eps):
        if self.passthrough:
            return eps
        t, b, l, r = self.conf.padding
        return eps[:, :, t:eps.shape[2] - b, l:eps.shape[3] - r]


class GlobalPoolingLayerImpl(LayerImpl):
    def forward(self, x, train=False):
        c = self.conf
        self._shape = x.shape
        dims = tuple(range(2, x.dim())) if c.poolingDimensions is None else tuple(c.poolingDimensions)
        self._dims = dims
        pt = c.poolingType
        m = self.maskArray
        if m is not None and x.dim() == 3:
            mm = m.reshape(x.shape[0], 1, x.shape[2]).to(x.dtype)
            if pt == "MAX":
                xm = x.masked_fill(mm == 0, float("-inf"))
                y, self._arg = xm.max(dim=2)
            elif pt in ("AVG", "SUM"):
                y = (x * mm).sum(2)
                if pt == "AVG":
                    self._cnt = mm.sum(2).clamp_min(1.0)
                    y = y / self._cnt
            else:
                y = ((x.abs() * mm) ** c.pnorm).sum(2) ** (1.0 / c.pnorm)
            self._mask3 = mm
        else:
            self._mask3 = None
            if pt == "MAX":
                flat = x.reshape(x.shape[0], x.shape[1], -1) if len(dims) == x.dim() - 2 else None
                if flat is not None:
                    y, self._arg = flat.max(dim=2)
                else:
                    y = x.amax(dim=dims)
                    self._arg = None
            elif pt == "AVG":
                y = x.mean(dim=dims)
            elif pt == "SUM":
                y = x.sum(dim=dims)
            else:
                y = (x.abs() ** c.pnorm).sum(dim=dims) ** (1.0 / c.pnorm)
        self.input, self._y = x, y
        if not c.collapseDimensions:
            for d in dims:
                y = y.unsqueeze(-1)
        return y

    def backward(self, eps):
        c, x = self.conf, self.input
        if not c.collapseDimensions:
            eps = eps.reshape(eps.shape[0], eps.shape[1])
        pt = c.poolingType
        e = eps.reshape(list(eps.shape) + [1] * (x.dim() - 2))
        if pt == "MAX":
            dx = torch.zeros(x.shape[0], x.shape[1], x[0, 0].numel(), dtype=eps.dtype, device=eps.device)
            dx.scatter_(2, self._arg.unsqueeze(-1), eps.unsqueeze(-1))
            return dx.reshape(x.shape)
        if pt == "AVG":
            if self._mask3 is not None:
                return (e / self._cnt.unsqueeze(-1)) * self._mask3
            n = x[0, 0].numel()
            return (e / n).expand(x.shape)
        if pt == "SUM":
            out = e.expand(x.shape)
            return out * self._mask3 if self._mask3 is not None else out
        p = c.pnorm
        y = self._y.reshape(e.shape)
        g = e * (x.abs() ** (p - 1)) * torch.sign(x) / (y ** (p - 1)).clamp_min(1e-12)
        return g * self._mask3 if self._mask3 is not None else g
