"""Gradient view object (reference: nn/gradient/DefaultGradient.java): per-variable gradient views
into the network's flat gradient vector."""
from __future__ import annotations

from .network_base import _strided_view


class Gradient:
    def __init__(self, net):
        self._net = net

    def gradient(self):
        return self._net._grads.view(1, -1)

    def gradientForVariable(self):
        out = {}
        for e in self._net._entries:
            out["%s_%s" % (self._net._layer_key(e.layer_idx), e.spec.key)] = _strided_view(self._net._grads, e.off, e.spec)
        return out

    def getGradientFor(self, name):
        return self.gradientForVariable()[name]

    def clear(self):
        self._net._grads.zero_()
