"""Runtime layer base classes.

DL4J layers implement ``activate(train, wsMgr)`` / ``backpropGradient(eps, wsMgr)`` with the math
done op-by-op from Java (reference: deeplearning4j-nn/.../nn/layers/**). Here every layer is ONE
fused forward and ONE fused backward on device-resident tensors; parameters and gradients are
views into the network's flat vectors (checkpoint ABI), optionally with compute-dtype (bf16)
replicas for the MFMA path.

Two flavours:
* ``LayerImpl``       — hand-written forward/backward (hot layers; HIP kernels underneath).
* ``AutogradLayerImpl`` — forward written once in PyTorch, backward derived by autograd locally
  inside the layer (the long tail; the analogue of the reference's SameDiff layers).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from ..conf import layers as C


class LayerCtx:
    """Per-network runtime context handed to every layer."""

    def __init__(self, dtype=torch.float32, cdtype=None, device="cpu", seed=12345, channels_last=None):
        self.dtype = dtype
        self.cdtype = cdtype or dtype
        self.device = torch.device(device)
        self.seed = seed
        self.channels_last = (self.device.type == "cuda") if channels_last is None else channels_last
        self.iteration = 0
        self.epoch = 0
        self.grads_zeroed = False        # set by the network once it clears the flat gradient vector before backward
        self.fused_bias_into_bn = set()  # names of conv layers whose bias is folded into the following BN

    @property
    def mixed(self):
        return self.cdtype != self.dtype


class LayerImpl:
    def __init__(self, conf: C.Layer, ctx: LayerCtx):
        self.conf, self.ctx = conf, ctx
        self.params: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        self.cparams: Dict[str, torch.Tensor] = {}
        self.input = None
        self.maskArray = None
        self.index = 0
        self.name = conf.layerName
        self._drop_mask = None

    # ---- parameters
    def param_specs(self):
        return self.conf.param_specs()

    def compute_replicas(self):
        """[(key, layout)] parameters wanting a compute-dtype replica; layout 'plain' | 'ohwi'."""
        return []

    def set_views(self, params, grads):
        self.params, self.grads = params, grads

    def w(self, key):
        """Parameter in compute dtype (replica in mixed mode, master view otherwise)."""
        t = self.cparams.get(key)
        if t is not None:
            return t
        p = self.params[key]
        if p.dtype != self.ctx.cdtype:
            return p.to(self.ctx.cdtype)
        return p

    def numParams(self):
        return sum(s.numel for s in self.param_specs())

    def paramTable(self):
        return self.params

    def getParam(self, key):
        return self.params[key]

    # ---- forward / backward
    def forward(self, x, train=False):
        raise NotImplementedError

    def backward(self, eps):
        raise NotImplementedError

    def activate(self, x, train=False):
        return self.forward(x, train)

    def backpropGradient(self, eps):
        return self.backward(eps)

    def clear(self):
        self.input = None
        self.maskArray = None
        self._drop_mask = None

    def setMaskArray(self, m):
        self.maskArray = m

    def getHelper(self):
        """The GPU helper object of this layer (reference: Layer.getHelper(); None when the layer has no HIP fast path here)."""
        from .helpers import helper_for
        return helper_for(self)

    def feedForwardMaskArray(self, mask, minibatch=None):
        self.maskArray = mask
        return mask

    def isPretrainLayer(self):
        return False

    def type(self):
        return "FEED_FORWARD"

    # ---- input dropout (DL4J applies idropout to the layer INPUT: BaseLayer.applyDropOutIfNecessary)
    def _dropout_in(self, x, train):
        d = getattr(self.conf, "idropout", None)
        self._drop_mask = None
        if not train or d is None:
            return x
        if isinstance(d, C.Dropout):
            p = d.pSchedule.valueAt(self.ctx.iteration, self.ctx.epoch) if d.pSchedule else d.p
            if p >= 1.0:
                return x
            m = (torch.rand_like(x, dtype=torch.float32) < p).to(x.dtype) / p
            self._drop_mask = m
            return x * m
        if isinstance(d, C.SpatialDropout):
            shp = list(x.shape[:2]) + [1] * (x.dim() - 2)
            m = (torch.rand(shp, device=x.device) < d.p).to(x.dtype) / d.p
            self._drop_mask = m
            return x * m
        if isinstance(d, C.GaussianDropout):
            std = (d.rate / (1.0 - d.rate)) ** 0.5
            m = 1.0 + std * torch.randn_like(x)
            self._drop_mask = m
            return x * m
        if isinstance(d, C.GaussianNoise):
            return x + d.stddev * torch.randn_like(x)
        if isinstance(d, C.AlphaDropout):
            alpha_p = d.alphaPrime
            p = d.p
            a = (p + alpha_p * alpha_p * p * (1 - p)) ** -0.5
            b = -a * (1 - p) * alpha_p
            keep = (torch.rand_like(x, dtype=torch.float32) < p).to(x.dtype)
            self._drop_mask = a * keep
            return a * (x * keep + alpha_p * (1 - keep)) + b
        return x

    def _dropout_back(self, eps):
        if self._drop_mask is not None:
            return eps * self._drop_mask
        return eps


class AutogradLayerImpl(LayerImpl):
    """Layer whose backward is derived by autograd from ``_forward``."""

    def _forward(self, x, P, train):
        raise NotImplementedError

    def forward(self, x, train=False):
        x = self._dropout_in(x, train)
        if not train:
            with torch.no_grad():
                return self._forward(x, {k: self.w(k) for k in self.params}, False)
        xin = x.detach()
        if xin.is_floating_point():
            xin.requires_grad_(True)
        leaves = {k: self.w(k).detach().requires_grad_(True) for k in self.params}
        with torch.enable_grad():
            y = self._forward(xin, leaves, True)
        self._ag = (xin, leaves, y)
        return y.detach()

    def backward(self, eps):
        xin, leaves, y = self._ag
        self._ag = None
        keys = list(leaves.keys())
        targets = ([xin] if xin.requires_grad else []) + [leaves[k] for k in keys]
        gs = torch.autograd.grad(y, targets, eps.to(y.dtype), allow_unused=True)
        off = 0
        dx = None
        if xin.requires_grad:
            dx = gs[0]
            off = 1
        for k, g in zip(keys, gs[off:]):
            if k in self.grads:
                if g is None:
                    self.grads[k].zero_()
                else:
                    self.grads[k].copy_(g.reshape(self.grads[k].shape))
        if dx is None:
            return None
        return self._dropout_back(dx)

    def clear(self):
        super().clear()
        self._ag = None
