"""Early stopping (reference: deeplearning4j-nn/.../earlystopping/**: EarlyStoppingConfiguration, EarlyStoppingResult,
trainer/{BaseEarlyStoppingTrainer,EarlyStoppingTrainer,EarlyStoppingGraphTrainer}, termination/*, scorecalc/*, saver/*)."""
from __future__ import annotations

import math
import os
import time
from typing import Dict, List, Optional

import torch


# ----------------------------------------------------------------------------- termination conditions
class EpochTerminationCondition:
    def initialize(self):
        pass

    def terminate(self, epochNum, score, minimize) -> bool:
        raise NotImplementedError


class IterationTerminationCondition:
    def initialize(self):
        pass

    def terminate(self, lastMiniBatchScore) -> bool:
        raise NotImplementedError


class MaxEpochsTerminationCondition(EpochTerminationCondition):
    def __init__(self, maxEpochs):
        if maxEpochs <= 0:
            raise ValueError("Max number of epochs must be >= 1")
        self.maxEpochs = int(maxEpochs)

    def terminate(self, epochNum, score, minimize):
        return epochNum + 1 >= self.maxEpochs

    def __str__(self):
        return "MaxEpochsTerminationCondition(%d)" % self.maxEpochs


class ScoreImprovementEpochTerminationCondition(EpochTerminationCondition):
    """Stop after N epochs without (at least minImprovement) improvement."""

    def __init__(self, maxEpochsWithNoImprovement, minImprovement=0.0):
        self.maxNoImprove, self.minImprovement = int(maxEpochsWithNoImprovement), float(minImprovement)
        self.initialize()

    def initialize(self):
        self.bestEpoch, self.bestScore = -1, float("nan")

    def terminate(self, epochNum, score, minimize):
        if self.bestEpoch == -1:
            self.bestEpoch, self.bestScore = epochNum, score
            return False
        improvement = (self.bestScore - score) if minimize else (score - self.bestScore)
        if improvement > self.minImprovement:
            self.bestScore, self.bestEpoch = score, epochNum
            return False
        return epochNum >= self.bestEpoch + self.maxNoImprove

    def __str__(self):
        return "ScoreImprovementEpochTerminationCondition(maxEpochsWithNoImprovement=%d, minImprovement=%s)" % (self.maxNoImprove, self.minImprovement)


class BestScoreEpochTerminationCondition(EpochTerminationCondition):
    """Stop once the score is at least as good as a target value."""

    def __init__(self, bestExpectedScore, lesserBetter=None):
        self.best = float(bestExpectedScore)

    def terminate(self, epochNum, score, minimize):
        return score < self.best if minimize else self.best < score

    def __str__(self):
        return "BestScoreEpochTerminationCondition(%s)" % self.best


class MaxTimeIterationTerminationCondition(IterationTerminationCondition):
    def __init__(self, maxTime, unit="seconds"):
        mult = {"milliseconds": 1e-3, "seconds": 1.0, "minutes": 60.0, "hours": 3600.0}[str(unit).lower()]
        self.maxSeconds = float(maxTime) * mult
        self.initialize()

    def initialize(self):
        self._start = time.time()

    def terminate(self, lastMiniBatchScore):
        return time.time() - self._start >= self.maxSeconds

    def __str__(self):
        return "MaxTimeIterationTerminationCondition(%ss)" % self.maxSeconds


class MaxScoreIterationTerminationCondition(IterationTerminationCondition):
    def __init__(self, maxScore):
        self.maxScore = float(maxScore)

    def terminate(self, s):
        return s > self.maxScore

    def __str__(self):
        return "MaxScoreIterationTerminationCondition(%s)" % self.maxScore


class InvalidScoreIterationTerminationCondition(IterationTerminationCondition):
    def terminate(self, s):
        return math.isnan(s) or math.isinf(s)

    def __str__(self):
        return "InvalidScoreIterationTerminationCondition()"


# ----------------------------------------------------------------------------- score calculators
class ScoreCalculator:
    def calculateScore(self, net) -> float:
        raise NotImplementedError

    def minimizeScore(self) -> bool:
        return True


def _reset(it):
    if hasattr(it, "reset"):
        it.reset()
    return it


class DataSetLossCalculator(ScoreCalculator):
    """Average (or summed) loss over a held-out iterator."""

    def __init__(self, iterator, average=True):
        self.iterator, self.average = iterator, average

    def calculateScore(self, net):
        tot, n = 0.0, 0
        for ds in _reset(self.iterator):
            mb = ds.numExamples()
            tot += float(net.score(ds, False)) * mb
            n += mb
        return tot / max(n, 1) if self.average else tot


DataSetLossCalculatorCG = DataSetLossCalculator


class ClassificationScoreCalculator(ScoreCalculator):
    def __init__(self, metric, iterator):
        self.metric, self.iterator = str(metric).upper(), iterator

    def calculateScore(self, net):
        e = net.evaluate(_reset(self.iterator))
        return {"ACCURACY": e.accuracy, "F1": e.f1, "PRECISION": e.precision, "RECALL": e.recall, "GMEASURE": e.gMeasure,
                "MCC": e.matthewsCorrelation}[self.metric]()

    def minimizeScore(self):
        return False


class RegressionScoreCalculator(ScoreCalculator):
    def __init__(self, metric, iterator):
        self.metric, self.iterator = str(metric).upper(), iterator

    def calculateScore(self, net):
        e = net.evaluateRegression(_reset(self.iterator))
        return {"MSE": e.averageMeanSquaredError, "MAE": e.averageMeanAbsoluteError, "RMSE": e.averagerootMeanSquaredError,
                "RSE": e.averagerelativeSquaredError, "PC": e.averagePearsonCorrelation, "R2": e.averageRSquared}[self.metric]()

    def minimizeScore(self):
        return self.metric in ("MSE", "MAE", "RMSE", "RSE")


class ROCScoreCalculator(ScoreCalculator):
    def __init__(self, type_, metric_or_iterator, iterator=None):
        if iterator is None:
            self.metric, self.iterator = "AUC", metric_or_iterator
        else:
            self.metric, self.iterator = str(metric_or_iterator).upper(), iterator
        self.type = str(type_).upper()

    def calculateScore(self, net):
        it = _reset(self.iterator)
        if self.type == "ROC":
            r = net.evaluateROC(it)
            return r.calculateAUC() if self.metric == "AUC" else r.calculateAUCPR()
        from .eval.evaluation import ROCBinary, ROCMultiClass
        r = net.doEvaluation(it, ROCBinary() if self.type == "BINARY" else ROCMultiClass())[0]
        return r.calculateAverageAUC() if self.metric == "AUC" else r.calculateAverageAUCPR()

    def minimizeScore(self):
        return False


class AutoencoderScoreCalculator(ScoreCalculator):
    """Reconstruction error of an AutoEncoder layer (layer 0 of the net) on a data set."""

    def __init__(self, metric, iterator):
        self.metric, self.iterator = str(metric).upper(), iterator

    def calculateScore(self, net):
        tot, n = 0.0, 0
        layer = net.getLayer(0)
        for ds in _reset(self.iterator):
            x = net._prep_features(ds.features)
            with torch.no_grad():
                rec = layer.reconstruct(x) if hasattr(layer, "reconstruct") else layer.forward(x, False)
            d = (rec - x).reshape(x.shape[0], -1)
            tot += float(((d * d).mean(1) if self.metric == "MSE" else d.abs().mean(1)).sum())
            n += x.shape[0]
        return tot / max(n, 1)


class VAEReconErrorScoreCalculator(AutoencoderScoreCalculator):
    pass


class VAEReconProbScoreCalculator(ScoreCalculator):
    def __init__(self, iterator, reconstructionProbNumSamples=1, logProb=True, average=True):
        self.iterator, self.n, self.logProb, self.average = iterator, reconstructionProbNumSamples, logProb, average

    def calculateScore(self, net):
        layer = net.getLayer(0)
        tot, cnt = 0.0, 0
        for ds in _reset(self.iterator):
            x = net._prep_features(ds.features)
            p = layer.reconstructionLogProbability(x, self.n)
            p = p if self.logProb else p.exp()
            tot += float(p.sum())
            cnt += x.shape[0]
        return tot / max(cnt, 1) if self.average else tot

    def minimizeScore(self):
        return False


# ----------------------------------------------------------------------------- savers
class InMemoryModelSaver:
    def __init__(self):
        self.best, self.latest = None, None

    def saveBestModel(self, net, score):
        self.best = net.clone()

    def saveLatestModel(self, net, score):
        self.latest = net.clone()

    def getBestModel(self):
        return self.best

    def getLatestModel(self):
        return self.latest


class LocalFileModelSaver:
    BEST, LATEST = "bestModel.bin", "latestModel.bin"

    def __init__(self, directory):
        self.directory = str(directory)
        os.makedirs(self.directory, exist_ok=True)

    def saveBestModel(self, net, score):
        net.save(os.path.join(self.directory, self.BEST), True)

    def saveLatestModel(self, net, score):
        net.save(os.path.join(self.directory, self.LATEST), True)

    def _load(self, name):
        from .utils.model_serializer import ModelGuesser
        p = os.path.join(self.directory, name)
        return ModelGuesser.loadModelGuess(p) if os.path.exists(p) else None

    def getBestModel(self):
        return self._load(self.BEST)

    def getLatestModel(self):
        return self._load(self.LATEST)


class LocalFileGraphSaver(LocalFileModelSaver):
    BEST, LATEST = "bestGraph.bin", "latestGraph.bin"


# ----------------------------------------------------------------------------- configuration / result
class EarlyStoppingConfiguration:
    def __init__(self, epochTerminationConditions=None, iterationTerminationConditions=None, scoreCalculator=None,
                 evaluateEveryNEpochs=1, modelSaver=None, saveLastModel=False):
        self.epochTerminationConditions = list(epochTerminationConditions or [])
        self.iterationTerminationConditions = list(iterationTerminationConditions or [])
        self.scoreCalculator = scoreCalculator
        self.evaluateEveryNEpochs = int(evaluateEveryNEpochs)
        self.modelSaver = modelSaver or InMemoryModelSaver()
        self.saveLastModel = bool(saveLastModel)

    def validate(self):
        if not self.epochTerminationConditions and not self.iterationTerminationConditions:
            raise ValueError("Cannot conduct early stopping without a termination condition (both Iteration and Epoch termination "
                             "conditions are null/empty)")

    class Builder:
        def __init__(self):
            self._kw = {}

        def epochTerminationConditions(self, *c):
            self._kw["epochTerminationConditions"] = list(c[0]) if len(c) == 1 and isinstance(c[0], (list, tuple)) else list(c)
            return self

        def iterationTerminationConditions(self, *c):
            self._kw["iterationTerminationConditions"] = list(c[0]) if len(c) == 1 and isinstance(c[0], (list, tuple)) else list(c)
            return self

        def scoreCalculator(self, s):
            self._kw["scoreCalculator"] = s
            return self

        def evaluateEveryNEpochs(self, n):
            self._kw["evaluateEveryNEpochs"] = n
            return self

        def modelSaver(self, s):
            self._kw["modelSaver"] = s
            return self

        def saveLastModel(self, b):
            self._kw["saveLastModel"] = b
            return self

        def build(self):
            return EarlyStoppingConfiguration(**self._kw)


class TerminationReason:
    Error, IterationTerminationCondition, EpochTerminationCondition = "Error", "IterationTerminationCondition", "EpochTerminationCondition"


class EarlyStoppingResult:
    def __init__(self, terminationReason, terminationDetails, scoreVsEpoch, bestModelEpoch, bestModelScore, totalEpochs, bestModel):
        self.terminationReason, self.terminationDetails = terminationReason, terminationDetails
        self.scoreVsEpoch: Dict[int, float] = scoreVsEpoch
        self.bestModelEpoch, self.bestModelScore, self.totalEpochs, self.bestModel = bestModelEpoch, bestModelScore, totalEpochs, bestModel

    def getBestModel(self):
        return self.bestModel

    getTerminationReason = lambda self: self.terminationReason
    getTerminationDetails = lambda self: self.terminationDetails
    getScoreVsEpoch = lambda self: self.scoreVsEpoch
    getBestModelEpoch = lambda self: self.bestModelEpoch
    getBestModelScore = lambda self: self.bestModelScore
    getTotalEpochs = lambda self: self.totalEpochs

    def __str__(self):
        return ("EarlyStoppingResult(terminationReason=%s,details=%s,bestModelEpoch=%s,bestModelScore=%s,totalEpochs=%s)" %
                (self.terminationReason, self.terminationDetails, self.bestModelEpoch, self.bestModelScore, self.totalEpochs))


class EarlyStoppingListener:
    def onStart(self, esConfig, net):
        pass

    def onEpoch(self, epochNum, score, esConfig, net):
        pass

    def onCompletion(self, esResult):
        pass


# ----------------------------------------------------------------------------- trainer
class BaseEarlyStoppingTrainer:
    """BaseEarlyStoppingTrainer.fit(): per-minibatch iteration conditions, per-epoch score + epoch conditions, best/last saving."""

    def __init__(self, esConfig: EarlyStoppingConfiguration, net, train, listener: Optional[EarlyStoppingListener] = None):
        self.esConfig, self.net, self.train, self.listener = esConfig, net, train, listener
        esConfig.validate()

    def setListener(self, l):
        self.listener = l

    def _fit_batch(self, ds):
        self.net.fit(ds)

    def fit(self) -> EarlyStoppingResult:
        c, net = self.esConfig, self.net
        for cond in c.epochTerminationConditions + c.iterationTerminationConditions:
            cond.initialize()
        if self.listener:
            self.listener.onStart(c, net)
        scoreVsEpoch: Dict[int, float] = {}
        bestScore, bestEpoch = None, -1
        epoch = 0
        minimize = c.scoreCalculator.minimizeScore() if c.scoreCalculator else True

        def result(reason, details, total):
            best = c.modelSaver.getBestModel()
            r = EarlyStoppingResult(reason, details, scoreVsEpoch, bestEpoch, bestScore if bestScore is not None else float("nan"), total, best)
            if self.listener:
                self.listener.onCompletion(r)
            return r

        while True:
            _reset(self.train)
            terminate, reason_cond = False, None
            try:
                for ds in self.train:
                    self._fit_batch(ds)
                    last = float(net.score())
                    for cond in c.iterationTerminationConditions:
                        if cond.terminate(last):
                            terminate, reason_cond = True, cond
                            break
                    if terminate:
                        break
            except Exception as e:                                   # reference: TerminationReason.Error with the exception text
                return result(TerminationReason.Error, repr(e), epoch)
            if terminate:
                if c.saveLastModel:
                    c.modelSaver.saveLatestModel(net, 0.0)
                return result(TerminationReason.IterationTerminationCondition, str(reason_cond), epoch)
            if hasattr(net, "incrementEpochCount") and not hasattr(self.train, "hasNext"):
                pass
            if (epoch == 0 and c.evaluateEveryNEpochs == 1) or epoch % c.evaluateEveryNEpochs == 0:
                score = c.scoreCalculator.calculateScore(net) if c.scoreCalculator else float(net.score())
                scoreVsEpoch[epoch] = score
                improved = bestScore is None or (score < bestScore if minimize else score > bestScore)
                if improved:
                    bestScore, bestEpoch = score, epoch
                    c.modelSaver.saveBestModel(net, score)
                if c.saveLastModel:
                    c.modelSaver.saveLatestModel(net, score)
                if self.listener:
                    self.listener.onEpoch(epoch, score, c, net)
                for cond in c.epochTerminationConditions:
                    if cond.terminate(epoch, score, minimize):
                        return result(TerminationReason.EpochTerminationCondition, str(cond), epoch + 1)
            else:
                for cond in c.epochTerminationConditions:
                    if isinstance(cond, MaxEpochsTerminationCondition) and cond.terminate(epoch, 0.0, minimize):
                        return result(TerminationReason.EpochTerminationCondition, str(cond), epoch + 1)
            epoch += 1


class EarlyStoppingTrainer(BaseEarlyStoppingTrainer):
    """For MultiLayerNetwork (accepts a configuration + init'ed network or a configuration to initialise)."""

    def __init__(self, esConfig, net, train, listener=None):
        from .nn.conf.configuration import MultiLayerConfiguration
        if isinstance(net, MultiLayerConfiguration):
            from .nn.multilayer import MultiLayerNetwork
            net = MultiLayerNetwork(net)
            net.init()
        super().__init__(esConfig, net, train, listener)


class EarlyStoppingGraphTrainer(BaseEarlyStoppingTrainer):
    def __init__(self, esConfig, net, train, listener=None):
        from .nn.conf.configuration import ComputationGraphConfiguration
        if isinstance(net, ComputationGraphConfiguration):
            from .nn.graph.computation_graph import ComputationGraph
            net = ComputationGraph(net)
            net.init()
        super().__init__(esConfig, net, train, listener)
