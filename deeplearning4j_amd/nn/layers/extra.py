"""Long-tail layers: forward written once in PyTorch, backward via local autograd
(AutogradLayerImpl). Reference implementations: deeplearning4j-nn/.../nn/layers/** and the SameDiff
layer definitions under nn/conf/layers/{SelfAttentionLayer,LearnedSelfAttentionLayer,
RecurrentAttentionLayer,CapsuleLayer,PrimaryCapsules,LocallyConnected1D/2D}.java.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from ...nd4j.activations import ActivationIdentity
from ..conf import layers as C
from ..conf.base import ConvolutionMode
from .base import AutogradLayerImpl, LayerImpl
from .core import OutputLayerImpl, LossLayerImpl


def _same_or(conf, sizes):
    """Return (padding tuple, extra F.pad list or None) for n-d convs."""
    mode = conf.convolutionMode or ConvolutionMode.Truncate
    if mode not in (ConvolutionMode.Same, ConvolutionMode.Causal):
        return tuple(conf.padding), None
    pads = []
    for i, sz in enumerate(sizes):
        if mode == ConvolutionMode.Causal:
            ke = conf.kernelSize[i] + (conf.kernelSize[i] - 1) * (conf.dilation[i] - 1)
            pads.append((ke - 1, 0))
        else:
            pads.append(C.same_pad(sz, conf.kernelSize[i], conf.stride[i], conf.dilation[i]))
    if all(p[0] == p[1] for p in pads):
        return tuple(p[0] for p in pads), None
    extra = []
    for p in reversed(pads):
        extra += [p[0], p[1]]
    return tuple(0 for _ in pads), extra


def _pad_pairs(conf, sizes):
    """(lo, hi) padding per spatial dim (Same mode may be asymmetric)."""
    mode = conf.convolutionMode or ConvolutionMode.Truncate
    if mode != ConvolutionMode.Same:
        return [(int(p), int(p)) for p in conf.padding]
    return [C.same_pad(sz, conf.kernelSize[i], conf.stride[i], conf.dilation[i]) for i, sz in enumerate(sizes)]


class Convolution1DLayerImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT"

    def _forward(self, x, P, train):
        c = self.conf
        pad, extra = _same_or(c, [x.shape[2]])
        if extra:
            x = F.pad(x, extra)
        w = P["W"][:, :, :, 0]
        b = P["b"].reshape(-1) if "b" in P else None
        z = F.conv1d(x, w, b, c.stride[0], pad[0], c.dilation[0])
        if self.maskArray is not None and z.shape[2] == self.maskArray.shape[1]:
            z = z * self.maskArray.unsqueeze(1).to(z.dtype)
        return c.activationFn.forward(z, train)


class Convolution3DImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        c = self.conf
        pad, extra = _same_or(c, x.shape[2:])
        if extra:
            x = F.pad(x, extra)
        b = P["b"].reshape(-1) if "b" in P else None
        return c.activationFn.forward(F.conv3d(x, P["W"], b, c.stride, pad, c.dilation), train)


class Deconvolution2DImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        c = self.conf
        b = P["b"].reshape(-1) if "b" in P else None
        if (c.convolutionMode or ConvolutionMode.Truncate) == ConvolutionMode.Same:
            z = F.conv_transpose2d(x, P["W"], b, c.stride, 0, 0, 1, c.dilation)
            oh, ow = x.shape[2] * c.stride[0], x.shape[3] * c.stride[1]
            th, tw = (z.shape[2] - oh) // 2, (z.shape[3] - ow) // 2
            z = z[:, :, th:th + oh, tw:tw + ow]
        else:
            z = F.conv_transpose2d(x, P["W"], b, c.stride, c.padding, 0, 1, c.dilation)
        return c.activationFn.forward(z, train)


class Deconvolution3DImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        c = self.conf
        b = P["b"].reshape(-1) if "b" in P else None
        if (c.convolutionMode or ConvolutionMode.Truncate) == ConvolutionMode.Same:
            z = F.conv_transpose3d(x, P["W"], b, c.stride, 0, 0, 1, c.dilation)
            tgt = [x.shape[2 + i] * c.stride[i] for i in range(3)]
            off = [(z.shape[2 + i] - tgt[i]) // 2 for i in range(3)]
            z = z[:, :, off[0]:off[0] + tgt[0], off[1]:off[1] + tgt[1], off[2]:off[2] + tgt[2]]
        else:
            z = F.conv_transpose3d(x, P["W"], b, c.stride, c.padding, 0, 1, c.dilation)
        return c.activationFn.forward(z, train)


class DepthwiseConvolution2DImpl(AutogradLayerImpl):
    """Depthwise stencil through ops/spatial.py (HIP kernels on the GPU in both directions)."""

    def _forward(self, x, P, train):
        from ...ops import spatial as SP
        c = self.conf
        # DL4J depthwise weights [kH, kW, nIn, depthMult] -> [nIn*mult, kH, kW]
        W = P["W"].permute(2, 3, 0, 1).reshape(c.nIn * c.depthMultiplier, c.kernelSize[0], c.kernelSize[1])
        b = P["b"].reshape(-1) if "b" in P else None
        return c.activationFn.forward(SP.depthwise_conv2d(x, W, b, c.stride, _pad_pairs(c, x.shape[2:]), c.dilation), train)


class SeparableConvolution2DImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        from ...ops import spatial as SP
        c = self.conf
        dw = P["W"].permute(1, 0, 2, 3).reshape(c.nIn * c.depthMultiplier, c.kernelSize[0], c.kernelSize[1])
        z = SP.depthwise_conv2d(x, dw, None, c.stride, _pad_pairs(c, x.shape[2:]), c.dilation)
        b = P["b"].reshape(-1) if "b" in P else None
        return c.activationFn.forward(F.conv2d(z, P["pW"].to(z.dtype), None if b is None else b.to(z.dtype)), train)


class Subsampling1DLayerImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT"

    def _forward(self, x, P, train):
        c = self.conf
        pad, extra = _same_or(c, [x.shape[2]])
        if extra:
            x = F.pad(x, extra, value=float("-inf") if c.poolingType == "MAX" else 0.0)
        k, s = c.kernelSize[0], c.stride[0]
        if c.poolingType == "MAX":
            return F.max_pool1d(x, k, s, pad[0], c.dilation[0])
        if c.poolingType == "AVG":
            return F.avg_pool1d(x, k, s, pad[0])
        if c.poolingType == "SUM":
            return F.avg_pool1d(x, k, s, pad[0]) * k
        p = c.pnorm or 2
        return (F.avg_pool1d(x.abs() ** p, k, s, pad[0]) * k) ** (1.0 / p)


class Subsampling3DLayerImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        from ...ops import spatial as SP
        c = self.conf
        return SP.pool3d(x, c.kernelSize, c.stride, _pad_pairs(c, x.shape[2:]), c.dilation, "max" if c.poolingType == "MAX" else "avg", True)


class UpsamplingImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT" if isinstance(self.conf, C.Upsampling1D) else "FEED_FORWARD"

    def _forward(self, x, P, train):
        s = self.conf.size
        if len(s) == 3 and x.dim() == 5:
            from ...ops import spatial as SP
            return SP.upsample3d(x, s)
        for d, r in enumerate(s):
            x = x.repeat_interleave(r, dim=2 + d)
        return x


class ZeroPaddingNDImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT" if isinstance(self.conf, C.ZeroPadding1DLayer) else "FEED_FORWARD"

    def _forward(self, x, P, train):
        p = list(self.conf.padding)
        pairs = [p[i:i + 2] for i in range(0, len(p), 2)]
        flat = []
        for a, b in reversed(pairs):
            flat += [a, b]
        return F.pad(x, flat)


class CroppingImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT" if isinstance(self.conf, C.Cropping1D) else "FEED_FORWARD"

    def _forward(self, x, P, train):
        c = list(self.conf.cropping)
        sl = [slice(None), slice(None)]
        for i in range(0, len(c), 2):
            d = 2 + i // 2
            sl.append(slice(c[i], x.shape[d] - c[i + 1]))
        return x[tuple(sl)]


class SpaceToDepthImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        b = self.conf.blockSize
        N, Cc, H, W = x.shape
        x = x.reshape(N, Cc, H // b, b, W // b, b).permute(0, 3, 5, 1, 2, 4)
        return x.reshape(N, Cc * b * b, H // b, W // b)


class SpaceToBatchImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        b, p = self.conf.blocks, self.conf.padding
        x = F.pad(x, (p[1][0], p[1][1], p[0][0], p[0][1]))
        N, Cc, H, W = x.shape
        x = x.reshape(N, Cc, H // b[0], b[0], W // b[1], b[1]).permute(3, 5, 0, 1, 2, 4)
        return x.reshape(N * b[0] * b[1], Cc, H // b[0], W // b[1])


class LocalResponseNormalizationImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        from ...ops import spatial as SP
        c = self.conf
        return SP.lrn(x, int(c.n), c.k, c.alpha, c.beta)


class PReLULayerImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        a = P["W"].reshape((1,) + tuple(P["W"].shape))
        return torch.where(x >= 0, x, x * a)


class ElementWiseMultiplicationLayerImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        return self.conf.activationFn.forward(x * P["W"].reshape(1, -1) + P["b"].reshape(1, -1), train)


class AutoEncoderImpl(AutogradLayerImpl):
    def isPretrainLayer(self):
        return True

    def _forward(self, x, P, train):
        return self.conf.activationFn.forward(x @ P["W"] + P["b"].reshape(1, -1), train)

    def pretrain_loss(self, x, P):
        c = self.conf
        xc = x
        if c.corruptionLevel > 0:
            xc = x * (torch.rand_like(x) >= c.corruptionLevel).to(x.dtype)
        h = c.activationFn.forward(xc @ P["W"] + P["b"].reshape(1, -1), True)
        z = h @ P["W"].t() + P["vb"].reshape(1, -1)
        return c.lossFn.computeScore(x, z, c.activationFn, None, False)


class VariationalAutoencoderImpl(AutogradLayerImpl):
    def isPretrainLayer(self):
        return True

    def _encode(self, x, P, train):
        c = self.conf
        h = x
        for i in range(len(c.encoderLayerSizes)):
            h = c.activationFn.forward(h @ P["e%dW" % i] + P["e%db" % i].reshape(1, -1), train)
        return h

    def _forward(self, x, P, train):
        h = self._encode(x, P, train)
        return self.conf.pzxActivationFn.forward(h @ P["pZXMeanW"] + P["pZXMeanb"].reshape(1, -1), train)

    def pretrain_loss(self, x, P):
        c = self.conf
        h = self._encode(x, P, True)
        mean = c.pzxActivationFn.forward(h @ P["pZXMeanW"] + P["pZXMeanb"].reshape(1, -1), True)
        logvar = c.pzxActivationFn.forward(h @ P["pZXLogStd2W"] + P["pZXLogStd2b"].reshape(1, -1), True)
        total = 0.0
        for _ in range(c.numSamples):
            z = mean + torch.exp(0.5 * logvar) * torch.randn_like(mean)
            d = z
            for i in range(len(c.decoderLayerSizes)):
                d = c.activationFn.forward(d @ P["d%dW" % i] + P["d%db" % i].reshape(1, -1), True)
            r = d @ P["pXZW"] + P["pXZb"].reshape(1, -1)
            total = total + self._neg_log_prob(x, r)
        kl = -0.5 * (1 + logvar - mean * mean - torch.exp(logvar)).sum()
        return total / c.numSamples + kl

    def _neg_log_prob(self, x, r):
        rd = self.conf.reconstructionDistribution
        t = rd.get("type", "gaussian")
        n = self.conf.nIn
        if t == "gaussian":
            mu, logs2 = r[:, :n], r[:, n:]
            return 0.5 * (math.log(2 * math.pi) + logs2 + (x - mu) ** 2 / torch.exp(logs2)).sum()
        if t == "bernoulli":
            return F.binary_cross_entropy_with_logits(r, x, reduction="sum")
        from ...nd4j.lossfunctions import get_loss
        return get_loss(rd.get("loss", "MSE")).computeScore(x, r, rd.get("activation", "IDENTITY"), None, False)

    def reconstructionLogProbability(self, x, numSamples=1):
        P = {k: self.w(k) for k in self.params}
        with torch.no_grad():
            old = self.conf.numSamples
            self.conf.numSamples = numSamples
            v = -self.pretrain_loss(x, P)
            self.conf.numSamples = old
        return v


class CenterLossOutputLayerImpl(OutputLayerImpl):
    def compute_score_sum(self):
        s = super().compute_score_sum()
        c = self.conf
        centers = self.params["cL"]
        lab_idx = self.labels.argmax(dim=1)
        diff = self.input.to(centers.dtype) - centers[lab_idx]
        self._center_diff, self._lab_idx = diff, lab_idx
        return s + 0.5 * c.lambda_ * (diff * diff).sum()

    def backward(self, eps=None):
        c = self.conf
        dx = super().backward(eps)
        diff, idx = self._center_diff, self._lab_idx
        # centre "gradient": moves centres towards the class mean with rate alpha (NoOp updater)
        g = self.grads["cL"]
        g.zero_()
        cnt = torch.zeros(g.shape[0], dtype=g.dtype, device=g.device).index_add_(0, idx, torch.ones_like(idx, dtype=g.dtype))
        g.index_add_(0, idx, -diff.to(g.dtype))
        g.mul_(c.alpha / (1.0 + cnt).unsqueeze(1))
        if dx is not None:
            dx = dx + c.lambda_ * diff.to(dx.dtype)
        return dx


class LocallyConnected2DImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        c = self.conf
        pad, extra = _same_or(c, x.shape[2:])
        if extra:
            x = F.pad(x, extra)
        cols = F.unfold(x, c.kernelSize, c.dilation, pad, c.stride)        # [N, fd, L]
        z = torch.einsum("nfl,lfo->nol", cols, P["W"])                       # [N, nOut, L]
        if "b" in P:
            z = z + P["b"].reshape(1, -1, 1)
        oh, ow = c.outputSize
        return c.activationFn.forward(z.reshape(x.shape[0], c.nOut, oh, ow), train)


class LocallyConnected1DImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT"

    def _forward(self, x, P, train):
        c = self.conf
        pad, extra = _same_or(c, [x.shape[2]])
        if extra:
            x = F.pad(x, extra)
        elif pad[0]:
            x = F.pad(x, (pad[0], pad[0]))
        cols = x.unfold(2, c.kernelSize[0], c.stride[0])                       # [N, C, L, k]
        cols = cols.permute(0, 2, 1, 3).reshape(x.shape[0], cols.shape[2], -1)  # [N, L, C*k]
        z = torch.einsum("nlf,lfo->nol", cols, P["W"])
        if "b" in P:
            z = z + P["b"].reshape(1, -1, 1)
        return c.activationFn.forward(z, train)


def _mha(q_in, k_in, v_in, P, nHeads, mask=None, project=True):
    """Multi-head dot-product attention on [mb, feat, T] tensors (DL4J layout). Returns [mb, nOut, Tq]."""
    from ...ops.attention import sdpa
    if not project:
        q, k, v = q_in.permute(0, 2, 1).unsqueeze(1), k_in.permute(0, 2, 1).unsqueeze(1), v_in.permute(0, 2, 1).unsqueeze(1)
        o = sdpa(q, k, v, mask)
        return o.squeeze(1).permute(0, 2, 1)
    q = torch.einsum("hsf,bft->bhts", P["Wq"], q_in)
    k = torch.einsum("hsf,bft->bhts", P["Wk"], k_in)
    v = torch.einsum("hsf,bft->bhts", P["Wv"], v_in)
    o = sdpa(q, k, v, mask)                                    # [b, h, Tq, hs]
    o = o.permute(0, 2, 1, 3).reshape(o.shape[0], o.shape[2], -1)    # [b, Tq, h*hs]
    return (o @ P["Wo"]).permute(0, 2, 1)


class SelfAttentionLayerImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT"

    def _forward(self, x, P, train):
        return _mha(x, x, x, P, self.conf.nHeads, self.maskArray, self.conf.projectInput)


class LearnedSelfAttentionLayerImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT"

    def _forward(self, x, P, train):
        q = P["Q"].expand(x.shape[0], -1, -1)
        return _mha(q, x, x, P, self.conf.nHeads, self.maskArray, self.conf.projectInput)

    def feedForwardMaskArray(self, mask, minibatch=None):
        self.maskArray = mask
        return None


class RecurrentAttentionLayerImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT"

    def _forward(self, x, P, train):
        c = self.conf
        mb, nIn, T = x.shape
        h = torch.zeros(mb, c.nOut, dtype=x.dtype, device=x.device)
        outs = []
        for t in range(T):
            att = _mha(h.unsqueeze(2), x, x, P, c.nHeads, self.maskArray, c.projectInput)[:, :, 0]
            z = x[:, :, t] @ P["W"] + att @ P["R"]
            if "b" in P:
                z = z + P["b"].reshape(1, -1)
            h = c.activationFn.forward(z, train)
            outs.append(h)
        return torch.stack(outs, dim=2)


class PrimaryCapsulesImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT"

    def _forward(self, x, P, train):
        c = self.conf
        b = P["b"].reshape(-1) if "b" in P else None
        z = F.conv2d(x, P["W"], b, c.stride, c.padding, c.dilation)
        if c.useRelu:
            z = F.leaky_relu(z, c.leak) if c.leak else torch.relu(z)
        z = z.reshape(x.shape[0], -1, c.capsuleDimensions)
        return _squash(z, 2)


def _squash(x, dim):
    n2 = (x * x).sum(dim, keepdim=True)
    return x * (n2 / (1.0 + n2)) / torch.sqrt(n2 + 1e-8)


class CapsuleLayerImpl(AutogradLayerImpl):
    def type(self):
        return "RECURRENT"

    def _forward(self, x, P, train):
        c = self.conf
        mb = x.shape[0]
        W = P["W"].reshape(c.inputCapsules, c.capsules, c.capsuleDimensions, c.inputCapsuleDimensions)
        u = torch.einsum("icdk,bik->bicd", W, x)                       # predictions
        b = torch.zeros(mb, c.inputCapsules, c.capsules, 1, dtype=x.dtype, device=x.device)
        v = None
        for r in range(c.routings):
            cc = torch.softmax(b, dim=2)
            s = (cc * u).sum(1)
            if "b" in P:
                s = s + P["b"].reshape(1, c.capsules, c.capsuleDimensions)
            v = _squash(s, 2)
            if r < c.routings - 1:
                b = b + (u * v.unsqueeze(1)).sum(3, keepdim=True)
        return v


class CapsuleStrengthLayerImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        return torch.sqrt((x * x).sum(2) + 1e-12)


class LambdaLayerImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        return self.conf.fn(x)


class SameDiffLayerImpl(AutogradLayerImpl):
    def _forward(self, x, P, train):
        return self.conf.defineLayer(x, P, self.maskArray)


class SameDiffOutputLayerImpl(LayerImpl):
    """User-defined output layer (SameDiffOutputLayer): activations + per-example loss from one torch function."""
    is_output = True

    def setLabels(self, labels, mask=None):
        self.labels, self.labelMask = labels, mask

    def forward(self, x, train=False):
        self.input = x
        with torch.no_grad():
            lab = getattr(self, "labels", None)
            if lab is None or lab.shape[0] != x.shape[0]:
                lab = None
            y, _ = self.conf.defineLayer(x, lab, {k: self.w(k) for k in self.params})
        return y

    def _loss(self):
        xin = self.input.detach().requires_grad_(True)
        leaves = {k: self.w(k).detach().requires_grad_(True) for k in self.params}
        with torch.enable_grad():
            _, per_ex = self.conf.defineLayer(xin, self.labels, leaves)
            per_ex = per_ex.reshape(per_ex.shape[0], -1).sum(1)
            if getattr(self, "labelMask", None) is not None:
                per_ex = per_ex * self.labelMask.reshape(per_ex.shape[0], -1)[:, 0].to(per_ex.dtype)
            total = per_ex.sum()
        return xin, leaves, per_ex, total

    def compute_score_sum(self):
        xin, leaves, per_ex, total = self._loss()
        keys = list(leaves)
        gs = torch.autograd.grad(total, [xin] + [leaves[k] for k in keys], allow_unused=True)
        self._gs = (gs[0], dict(zip(keys, gs[1:])))
        return total.detach()

    def score_array(self):
        return self._loss()[2].detach()

    def backward(self, eps=None):
        dx, dps = self._gs
        for k, g in dps.items():
            if k in self.grads:
                if g is None:
                    self.grads[k].zero_()
                else:
                    self.grads[k].copy_(g.reshape(self.grads[k].shape))
        return dx


class OCNNOutputLayerImpl(LayerImpl):
    """One-class NN output (conf/ocnn/OCNNOutputLayer.java): score = 0.5|W|^2+0.5|V|^2 + 1/nu*mean(relu(r - y)) - r."""
    is_output = True

    def setLabels(self, labels, mask=None):
        self.labels, self.labelMask = labels, mask

    def forward(self, x, train=False):
        self.input = x
        act = self.conf.activationFn
        self._h = act.forward(x @ self.w("W"), train)
        return self._h @ self.w("V")

    def compute_score_sum(self):
        c = self.conf
        x = self.input.detach()
        W = self.w("W").detach().requires_grad_(True)
        V = self.w("V").detach().requires_grad_(True)
        r = self.params["R"].detach().requires_grad_(True)
        xin = x.requires_grad_(True)
        with torch.enable_grad():
            y = c.activationFn.forward(xin @ W, True) @ V
            loss = (0.5 * (W * W).sum() + 0.5 * (V * V).sum() - r.sum()) * x.shape[0] \
                + (1.0 / c.nu) * torch.relu(r - y).sum()
        gs = torch.autograd.grad(loss, [xin, W, V, r])
        self._gs = gs
        return loss.detach()

    def score_array(self):
        return self.forward(self.input).reshape(-1)

    def backward(self, eps=None):
        dx, dW, dV, dr = self._gs
        self.grads["W"].copy_(dW); self.grads["V"].copy_(dV); self.grads["R"].copy_(dr.reshape(1, 1))
        return dx


class Yolo2OutputLayerImpl(LayerImpl):
    """YOLOv2 loss (objdetect/Yolo2OutputLayer.java). Labels [mb, 4+C, H, W] (x1,y1,x2,y2 in grid units + one-hot)."""
    is_output = True

    def setLabels(self, labels, mask=None):
        self.labels, self.labelMask = labels, mask

    def _split(self, x):
        bb = torch.as_tensor(self.conf.boundingBoxes, dtype=x.dtype, device=x.device)  # [B,2] (w,h) priors
        B = bb.shape[0]
        mb, ch, H, W = x.shape
        x5 = x.reshape(mb, B, ch // B, H, W)
        xy = torch.sigmoid(x5[:, :, 0:2])
        wh = torch.exp(x5[:, :, 2:4]) * bb.reshape(1, B, 2, 1, 1)
        conf = torch.sigmoid(x5[:, :, 4:5])
        cls = torch.softmax(x5[:, :, 5:], dim=2)
        return xy, wh, conf, cls

    def forward(self, x, train=False):
        self.input = x
        xy, wh, conf, cls = self._split(x)
        return torch.cat([xy, wh, conf, cls], dim=2).reshape(x.shape)

    def _loss(self, x):
        c = self.conf
        lab = self.labels.to(x.dtype)
        xy, wh, conf, cls = self._split(x)
        mb, B = xy.shape[0], xy.shape[1]
        H, W = x.shape[2], x.shape[3]
        tl, br = lab[:, 0:2], lab[:, 2:4]
        obj = ((br - tl).abs().sum(1, keepdim=True) > 0).to(x.dtype)         # [mb,1,H,W]
        centre = 0.5 * (tl + br)
        gx = torch.arange(W, device=x.device, dtype=x.dtype).reshape(1, 1, 1, W)
        gy = torch.arange(H, device=x.device, dtype=x.dtype).reshape(1, 1, H, 1)
        t_xy = torch.stack([centre[:, 0:1] - gx, centre[:, 1:2] - gy], dim=2).reshape(mb, 1, 2, H, W)
        t_wh = (br - tl).reshape(mb, 1, 2, H, W)
        # IOU between each predicted box and the label box (both centred in the same cell frame)
        p_tl, p_br = xy - 0.5 * wh, xy + 0.5 * wh
        l_tl, l_br = t_xy - 0.5 * t_wh, t_xy + 0.5 * t_wh
        inter = (torch.minimum(p_br, l_br) - torch.maximum(p_tl, l_tl)).clamp_min(0).prod(2)
        union = wh.prod(2) + t_wh.prod(2) - inter
        iou = (inter / union.clamp_min(1e-9)).detach()                        # [mb,B,H,W]
        best = (iou == iou.max(dim=1, keepdim=True)[0]).to(x.dtype) * obj
        resp = best.unsqueeze(2)
        pos = ((xy - t_xy) ** 2 * resp).sum() + ((torch.sqrt(wh) - torch.sqrt(t_wh.clamp_min(1e-9))) ** 2 * resp).sum()
        conf_l = ((conf.squeeze(2) - iou) ** 2 * best).sum() + c.lambdaNoObj * ((conf.squeeze(2)) ** 2 * (1 - best)).sum()
        cls_l = ((cls - lab[:, 4:].unsqueeze(1)) ** 2 * resp).sum()
        return c.lambdaCoord * pos + conf_l + cls_l

    def compute_score_sum(self):
        x = self.input.detach().requires_grad_(True)
        with torch.enable_grad():
            l = self._loss(x)
        self._dx = torch.autograd.grad(l, x)[0]
        return l.detach()

    def backward(self, eps=None):
        return self._dx
