"""Solver / optimizer objects of the DL4J training loop (reference: optimize/Solver.java, optimize/solvers/{BaseOptimizer,
StochasticGradientDescent}.java:50-76, optimize/api/ConvexOptimizer.java).

In the reference ``fit`` builds a ``Solver`` whose ``StochasticGradientDescent.optimize`` runs gradientAndScore ->
updateGradientAccordingToParams (MultiLayerUpdater -> UpdaterBlocks) -> stepFunction. Here the same sequence is the network's
``computeGradientAndScore`` followed by the single fused segmented-updater launch (``_apply_update``); these classes expose that
sequence under the reference's names so code that drives a Solver directly keeps working.
"""
from __future__ import annotations


class ConvexOptimizer:
    """optimize/api/ConvexOptimizer.java"""

    def optimize(self, *a, **k):
        raise NotImplementedError

    def score(self):
        raise NotImplementedError


class NegativeDefaultStepFunction:
    """params -= update (optimize/stepfunctions/NegativeDefaultStepFunction.java); fused into the updater kernel on the GPU."""

    def step(self, params, update, step=1.0):
        params.sub_(update * step)
        return params


class BaseOptimizer(ConvexOptimizer):
    def __init__(self, model, conf=None, stepFunction=None, listeners=()):
        self.model, self.conf = model, conf if conf is not None else getattr(model, "conf", None)
        self.stepFunction = stepFunction or NegativeDefaultStepFunction()
        self.listeners = list(listeners)
        self._score = float("nan")

    def gradientAndScore(self, features=None, labels=None, featuresMask=None, labelsMask=None):
        """BaseOptimizer.gradientAndScore (:153-169): forward + backward into the flat gradient vector; returns (gradient, score)."""
        m = self.model
        if features is not None:
            if hasattr(m, "topologicalOrder"):
                m.computeGradientAndScore(features if isinstance(features, (list, tuple)) else [features],
                                          labels if isinstance(labels, (list, tuple)) else [labels])
            else:
                m.computeGradientAndScore(features, labels, featuresMask, labelsMask)
        else:
            m.computeGradientAndScore()
        self._score = float(m.score())
        return m.gradient(), self._score

    def updateGradientAccordingToParams(self, batchSize):
        """:214-233 — updater blocks (one fused launch) + step function; advances the iteration counter like fit does."""
        m = self.model
        m._apply_update(int(batchSize))
        m.iterationCount += 1

    def score(self):
        return self._score

    def getUpdater(self):
        from ..nn.updater import MultiLayerUpdater, ComputationGraphUpdater
        return ComputationGraphUpdater(self.model) if hasattr(self.model, "topologicalOrder") else MultiLayerUpdater(self.model)


class StochasticGradientDescent(BaseOptimizer):
    """One optimize() call = one minibatch iteration (StochasticGradientDescent.optimize :50-76)."""

    def optimize(self, features=None, labels=None, featuresMask=None, labelsMask=None):
        first = features[0] if isinstance(features, (list, tuple)) else features
        n = int(first.shape[0]) if first is not None else int(getattr(self.model, "_last_batch", 1) or 1)
        self.gradientAndScore(features, labels, featuresMask, labelsMask)
        self.updateGradientAccordingToParams(n)
        for l in self.listeners:
            if hasattr(l, "iterationDone"):
                l.iterationDone(self.model, self.model.iterationCount, self.model.epochCount)
        return True


class Solver:
    """optimize/Solver.java: ``new Solver.Builder().configure(conf).model(net).listeners(..).build().optimize(...)``."""

    def __init__(self, model=None, conf=None, listeners=()):
        self.model, self.conf, self.listeners = model, conf, list(listeners)
        self._opt = None

    class Builder:
        def __init__(self):
            self._m = self._c = None
            self._l = []

        def configure(self, conf):
            self._c = conf
            return self

        def model(self, m):
            self._m = m
            return self

        def listener(self, *ls):
            self._l += list(ls)
            return self

        listeners = listener

        def build(self):
            return Solver(self._m, self._c, self._l)

    def getOptimizer(self):
        if self._opt is None:
            self._opt = StochasticGradientDescent(self.model, self.conf, listeners=self.listeners)
        return self._opt

    def optimize(self, features=None, labels=None, featuresMask=None, labelsMask=None):
        return self.getOptimizer().optimize(features, labels, featuresMask, labelsMask)

    def setListeners(self, ls):
        self.listeners = list(ls)
        if self._opt is not None:
            self._opt.listeners = self.listeners
