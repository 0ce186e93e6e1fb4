"""Updater objects over the flat parameter vector (reference: nn/updater/{UpdaterBlock (:131-166), BaseMultiLayerUpdater,
MultiLayerUpdater, graph/ComputationGraphUpdater}.java).

The reference fragments the flat vector into UpdaterBlocks (contiguous parameters sharing an updater configuration) and applies one
fused updater op per block. The networks here keep exactly that block structure as the segment table of ONE kernel launch
(ops/updater.py SegmentedUpdater, native/csrc/updaters.hip); these classes are the reference-named views of it.
"""
from __future__ import annotations

import torch


class UpdaterBlock:
    """A contiguous [paramOffsetStart, paramOffsetEnd) range of the flat vector with one updater configuration and its state view."""

    def __init__(self, net, index, seg, updater):
        from ..ops import updater as U
        self._net, self.index = net, index
        self.paramOffsetStart, self.paramOffsetEnd = int(seg.off), int(seg.off + seg.length)
        self.updaterViewOffsetStart = int(seg.state_off)
        self.updaterViewOffsetEnd = int(seg.state_off + U.STATE_MULT[seg.kind] * seg.length)
        self.gradientUpdater = updater

    def getUpdaterView(self):
        return self._net._updater_state[self.updaterViewOffsetStart:self.updaterViewOffsetEnd]

    def getGradientView(self):
        return self._net._grads.reshape(-1)[self.paramOffsetStart:self.paramOffsetEnd]

    def getParamView(self):
        return self._net._params.reshape(-1)[self.paramOffsetStart:self.paramOffsetEnd]

    def layersAndVariablesInBlock(self):
        out = []
        for e in self._net._entries:
            if self.paramOffsetStart <= e.off < self.paramOffsetEnd:
                out.append((getattr(e, "layer", None), e.spec.key))
        return out

    def __repr__(self):
        return "UpdaterBlock(%d: params [%d, %d), %s)" % (self.index, self.paramOffsetStart, self.paramOffsetEnd,
                                                         type(self.gradientUpdater).__name__)


class BaseMultiLayerUpdater:
    def __init__(self, network):
        self.network = network

    def getUpdaterBlocks(self):
        n = self.network
        return [UpdaterBlock(n, k, s, u) for k, (s, u) in enumerate(zip(n._updater.segments, n._seg_updaters))]

    def getStateViewArray(self):
        return self.network.getUpdaterState()

    def setStateViewArray(self, view, initialize=False):
        self.network.setUpdaterState(view)

    def getFlattenedGradientsView(self):
        return self.network._grads

    def getParams(self):
        return self.network.params()

    def preApply(self):
        """Gradient normalization / clipping (BaseMultiLayerUpdater.preApply)."""
        self.network._pre_apply_gradient_normalization()

    def update(self, gradient=None, iteration=None, epoch=None, batchSize=1):
        """Apply every updater block to the flat gradient (in place: the gradient vector becomes the update and the parameters are
        stepped, as MultiLayerUpdater.update + stepFunction do); one kernel launch on the GPU."""
        n = self.network
        if gradient is not None and gradient is not n._grads:
            n._grads.reshape(-1).copy_(torch.as_tensor(gradient).reshape(-1).to(n._grads.dtype))
        if iteration is not None:
            n.iterationCount = int(iteration)
            n._updater.set_iteration(int(iteration))
        if epoch is not None:
            n.epochCount = int(epoch)
        n._apply_update(int(batchSize))

    def __eq__(self, o):
        return type(self) is type(o) and self.network is o.network

    def __hash__(self):
        return id(self.network)


class MultiLayerUpdater(BaseMultiLayerUpdater):
    pass


class ComputationGraphUpdater(BaseMultiLayerUpdater):
    pass


class UpdaterCreator:
    """nn/updater/UpdaterCreator.java"""

    @staticmethod
    def getUpdater(network):
        return ComputationGraphUpdater(network) if hasattr(network, "topologicalOrder") else MultiLayerUpdater(network)
