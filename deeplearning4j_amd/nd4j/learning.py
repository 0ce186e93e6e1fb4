"""IUpdater configs, LR schedules and regularization (reference: nd4j-api/.../linalg/learning/config/*,
linalg/schedule/*, linalg/learning/regularization/*). JSON shapes follow the Jackson output of the
Java classes ("@class" discriminators) so configuration.json round-trips.

The math itself lives in the fused segmented updater (ops/updater.py + native/csrc/updaters.hip);
``GradientUpdater`` below is the single-block API-parity wrapper (AdamUpdater.applyUpdater etc.).
"""
from __future__ import annotations

import math

import torch

from ..ops import updater as U

_CFG = "org.nd4j.linalg.learning.config."
_SCH = "org.nd4j.linalg.schedule."
_REG = "org.nd4j.linalg.learning.regularization."


# ------------------------------------------------------------------ schedules
class ISchedule:
    JAVA_CLASS = None

    def valueAt(self, iteration, epoch):
        raise NotImplementedError

    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        d.update({k: v for k, v in self.__dict__.items() if not k.startswith("_")})
        return d

    def __eq__(self, o):
        return type(self) is type(o) and self.__dict__ == o.__dict__

    def __hash__(self):
        return hash(type(self).__name__)


class ScheduleType:
    ITERATION = "ITERATION"
    EPOCH = "EPOCH"


def _t(stype, it, ep):
    return it if stype == ScheduleType.ITERATION else ep


class FixedSchedule(ISchedule):
    JAVA_CLASS = _SCH + "FixedSchedule"

    def __init__(self, value):
        self.value = float(value)

    def valueAt(self, iteration, epoch):
        return self.value


class ExponentialSchedule(ISchedule):
    JAVA_CLASS = _SCH + "ExponentialSchedule"

    def __init__(self, scheduleType, initialValue, gamma):
        self.scheduleType, self.initialValue, self.gamma = scheduleType, float(initialValue), float(gamma)

    def valueAt(self, iteration, epoch):
        return self.initialValue * self.gamma ** _t(self.scheduleType, iteration, epoch)


class InverseSchedule(ISchedule):
    JAVA_CLASS = _SCH + "InverseSchedule"

    def __init__(self, scheduleType, initialValue, gamma, power):
        self.scheduleType, self.initialValue, self.gamma, self.power = scheduleType, float(initialValue), float(gamma), float(power)

    def valueAt(self, iteration, epoch):
        return self.initialValue / (1 + self.gamma * _t(self.scheduleType, iteration, epoch)) ** self.power


class PolySchedule(ISchedule):
    JAVA_CLASS = _SCH + "PolySchedule"

    def __init__(self, scheduleType, initialValue, power, maxIter):
        self.scheduleType, self.initialValue, self.power, self.maxIter = scheduleType, float(initialValue), float(power), int(maxIter)

    def valueAt(self, iteration, epoch):
        return self.initialValue * (1 + (1.0 * _t(self.scheduleType, iteration, epoch)) / self.maxIter) ** (-self.power)


class SigmoidSchedule(ISchedule):
    JAVA_CLASS = _SCH + "SigmoidSchedule"

    def __init__(self, scheduleType, initialValue, gamma, stepSize):
        self.scheduleType, self.initialValue, self.gamma, self.stepSize = scheduleType, float(initialValue), float(gamma), int(stepSize)

    def valueAt(self, iteration, epoch):
        return self.initialValue / (1 + math.exp(-self.gamma * (_t(self.scheduleType, iteration, epoch) - self.stepSize)))


class StepSchedule(ISchedule):
    JAVA_CLASS = _SCH + "StepSchedule"

    def __init__(self, scheduleType, initialValue, decayRate, step):
        self.scheduleType, self.initialValue, self.decayRate, self.step = scheduleType, float(initialValue), float(decayRate), float(step)

    def valueAt(self, iteration, epoch):
        return self.initialValue * self.decayRate ** math.floor(_t(self.scheduleType, iteration, epoch) / self.step)


class MapSchedule(ISchedule):
    JAVA_CLASS = _SCH + "MapSchedule"

    def __init__(self, scheduleType, values):
        self.scheduleType = scheduleType
        self.values = {int(k): float(v) for k, v in values.items()}
        if 0 not in self.values:
            raise ValueError("MapSchedule must contain a value for position 0")

    def valueAt(self, iteration, epoch):
        t = _t(self.scheduleType, iteration, epoch)
        keys = sorted(k for k in self.values if k <= t)
        return self.values[keys[-1]]


class CycleSchedule(ISchedule):
    JAVA_CLASS = _SCH + "CycleSchedule"

    def __init__(self, scheduleType, initialLearningRate, maxLearningRate, cycleLength, annealingLength, annealingDecay):
        self.scheduleType = scheduleType
        self.initialLearningRate, self.maxLearningRate = float(initialLearningRate), float(maxLearningRate)
        self.cycleLength, self.annealingLength, self.annealingDecay = int(cycleLength), int(annealingLength), float(annealingDecay)

    def valueAt(self, iteration, epoch):
        t = _t(self.scheduleType, iteration, epoch)
        step_size = (self.cycleLength - self.annealingLength) // 2
        pos = t % self.cycleLength
        if pos < step_size:
            return self.initialLearningRate + (self.maxLearningRate - self.initialLearningRate) * pos / step_size
        if pos < 2 * step_size:
            return self.maxLearningRate - (self.maxLearningRate - self.initialLearningRate) * (pos - step_size) / step_size
        return self.initialLearningRate * self.annealingDecay ** (pos - 2 * step_size)


class RampSchedule(ISchedule):
    JAVA_CLASS = _SCH + "RampSchedule"

    def __init__(self, baseSchedule, numIter):
        self.baseSchedule, self.numIter = baseSchedule, int(numIter)

    def valueAt(self, iteration, epoch):
        b = self.baseSchedule.valueAt(iteration, epoch)
        return b if iteration >= self.numIter - 1 else b * (iteration + 1) / self.numIter

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "baseSchedule": self.baseSchedule.to_json(), "numIter": self.numIter}


_SCHEDULES = {c.JAVA_CLASS: c for c in (FixedSchedule, ExponentialSchedule, InverseSchedule, PolySchedule,
                                        SigmoidSchedule, StepSchedule, MapSchedule, CycleSchedule, RampSchedule)}


def schedule_from_json(d):
    if d is None:
        return None
    cls = _SCHEDULES[d["@class"]]
    kw = {k: v for k, v in d.items() if k != "@class"}
    if cls is RampSchedule:
        kw["baseSchedule"] = schedule_from_json(kw["baseSchedule"])
    return cls(**kw)


# ------------------------------------------------------------------ updaters
class IUpdater:
    JAVA_CLASS = None
    KIND = "NoOp"
    FIELDS = ()

    def stateSize(self, numParams):
        return U.STATE_MULT[U.KIND[self.KIND]] * numParams

    def hasLearningRate(self):
        return hasattr(self, "learningRate")

    def getLearningRate(self, iteration=0, epoch=0):
        sch = getattr(self, "learningRateSchedule", None)
        if sch is not None:
            return sch.valueAt(iteration, epoch)
        return getattr(self, "learningRate", float("nan"))

    def setLrAndSchedule(self, lr, schedule):
        self.learningRate, self.learningRateSchedule = lr, schedule

    def state_init_value(self):
        return 0.0

    def segment_hparams(self, iteration=0, epoch=0):
        """(lr, a, b, eps) in the fused kernel's convention."""
        return (0.0, 0.0, 0.0, 0.0)

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        for f in self.FIELDS:
            v = getattr(self, f)
            d[f] = v.to_json() if isinstance(v, ISchedule) else v
        return d

    def __eq__(self, o):
        return type(self) is type(o) and all(getattr(self, f) == getattr(o, f) for f in self.FIELDS)

    def __hash__(self):
        return hash(type(self).__name__)

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join("%s=%r" % (f, getattr(self, f)) for f in self.FIELDS))

    def instantiate(self, stateView, initialize=True):
        return GradientUpdater(self, stateView, initialize)


class NoOp(IUpdater):
    JAVA_CLASS, KIND = _CFG + "NoOp", "NoOp"


class Sgd(IUpdater):
    JAVA_CLASS, KIND, FIELDS = _CFG + "Sgd", "Sgd", ("learningRate", "learningRateSchedule")
    DEFAULT_SGD_LR = 1e-3

    def __init__(self, learningRate=1e-3, learningRateSchedule=None):
        if isinstance(learningRate, ISchedule):
            learningRate, learningRateSchedule = 1e-3, learningRate
        self.learningRate, self.learningRateSchedule = float(learningRate), learningRateSchedule

    def segment_hparams(self, it=0, ep=0):
        return (self.getLearningRate(it, ep), 0.0, 0.0, 0.0)


class Nesterovs(IUpdater):
    JAVA_CLASS, KIND = _CFG + "Nesterovs", "Nesterovs"
    FIELDS = ("learningRate", "learningRateSchedule", "momentum", "momentumISchedule")

    def __init__(self, learningRate=0.1, momentum=0.9, learningRateSchedule=None, momentumISchedule=None):
        if isinstance(learningRate, ISchedule):
            learningRate, learningRateSchedule = 0.1, learningRate
        self.learningRate, self.momentum = float(learningRate), float(momentum)
        self.learningRateSchedule, self.momentumISchedule = learningRateSchedule, momentumISchedule

    def segment_hparams(self, it=0, ep=0):
        mu = self.momentumISchedule.valueAt(it, ep) if self.momentumISchedule else self.momentum
        return (self.getLearningRate(it, ep), mu, 0.0, 0.0)


class _AdamLike(IUpdater):
    FIELDS = ("learningRate", "learningRateSchedule", "beta1", "beta2", "epsilon")

    def __init__(self, learningRate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8, learningRateSchedule=None):
        if isinstance(learningRate, ISchedule):
            learningRate, learningRateSchedule = 1e-3, learningRate
        self.learningRate, self.beta1, self.beta2, self.epsilon = float(learningRate), float(beta1), float(beta2), float(epsilon)
        self.learningRateSchedule = learningRateSchedule

    def segment_hparams(self, it=0, ep=0):
        return (self.getLearningRate(it, ep), self.beta1, self.beta2, self.epsilon)


class Adam(_AdamLike):
    JAVA_CLASS, KIND = _CFG + "Adam", "Adam"


class AdaMax(_AdamLike):
    JAVA_CLASS, KIND = _CFG + "AdaMax", "AdaMax"


class Nadam(_AdamLike):
    JAVA_CLASS, KIND = _CFG + "Nadam", "Nadam"


class AMSGrad(_AdamLike):
    JAVA_CLASS, KIND = _CFG + "AMSGrad", "AMSGrad"


class AdaBelief(_AdamLike):
    JAVA_CLASS, KIND = _CFG + "AdaBelief", "AdaBelief"

    def __init__(self, learningRate=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-14, learningRateSchedule=None):
        super().__init__(learningRate, beta1, beta2, epsilon, learningRateSchedule)


class RmsProp(IUpdater):
    JAVA_CLASS, KIND = _CFG + "RmsProp", "RmsProp"
    FIELDS = ("learningRate", "learningRateSchedule", "rmsDecay", "epsilon")

    def __init__(self, learningRate=0.1, rmsDecay=0.95, epsilon=1e-8, learningRateSchedule=None):
        if isinstance(learningRate, ISchedule):
            learningRate, learningRateSchedule = 0.1, learningRate
        self.learningRate, self.rmsDecay, self.epsilon = float(learningRate), float(rmsDecay), float(epsilon)
        self.learningRateSchedule = learningRateSchedule

    def state_init_value(self):
        return self.epsilon  # RmsPropUpdater.setStateViewArray assigns epsilon (RmsPropUpdater.java:64)

    def segment_hparams(self, it=0, ep=0):
        return (self.getLearningRate(it, ep), self.rmsDecay, 0.0, self.epsilon)


class AdaGrad(IUpdater):
    JAVA_CLASS, KIND = _CFG + "AdaGrad", "AdaGrad"
    FIELDS = ("learningRate", "learningRateSchedule", "epsilon")

    def __init__(self, learningRate=0.1, epsilon=1e-6, learningRateSchedule=None):
        if isinstance(learningRate, ISchedule):
            learningRate, learningRateSchedule = 0.1, learningRate
        self.learningRate, self.epsilon, self.learningRateSchedule = float(learningRate), float(epsilon), learningRateSchedule

    def state_init_value(self):
        return self.epsilon  # AdaGradUpdater.java:69

    def segment_hparams(self, it=0, ep=0):
        return (self.getLearningRate(it, ep), 0.0, 0.0, self.epsilon)


class AdaDelta(IUpdater):
    JAVA_CLASS, KIND, FIELDS = _CFG + "AdaDelta", "AdaDelta", ("rho", "epsilon")

    def __init__(self, rho=0.95, epsilon=1e-6):
        self.rho, self.epsilon = float(rho), float(epsilon)

    def hasLearningRate(self):
        return False

    def segment_hparams(self, it=0, ep=0):
        return (0.0, self.rho, 0.0, self.epsilon)


_UPDATERS = {c.JAVA_CLASS: c for c in (NoOp, Sgd, Nesterovs, Adam, AdaMax, Nadam, AMSGrad, AdaBelief, RmsProp, AdaGrad, AdaDelta)}


def updater_from_json(d):
    if d is None:
        return None
    cls = _UPDATERS[d["@class"]]
    kw = {}
    for k, v in d.items():
        if k == "@class":
            continue
        if isinstance(v, dict) and "@class" in v:
            v = schedule_from_json(v)
        kw[k] = v
    return cls(**kw)


class GradientUpdater:
    """Stateful single-block updater (API parity with nd4j GradientUpdater.applyUpdater):
    transforms ``gradient`` in place into the update, using/advancing ``stateView``."""

    def __init__(self, config: IUpdater, stateView, initialize=True):
        self.config, self.state = config, stateView
        if initialize and stateView is not None and stateView.numel():
            stateView.fill_(config.state_init_value())

    def applyUpdater(self, gradient, iteration=0, epoch=0):
        g = gradient.reshape(-1)
        lr, a, b, eps = self.config.segment_hparams(iteration, epoch)
        seg = U.Segment(0, g.numel(), 0, U.KIND[self.config.KIND], lr, a, b, eps, divide_by_batch=False,
                        update_only=True)
        if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous():
            su = U.SegmentedUpdater([seg], g.device)
            su.set_iteration(iteration)
            st = self.state if self.state is not None else g
            su.step(g, g, st, 1)
        else:
            st = self.state if self.state is not None else g
            U.apply_segment_reference(seg, g, g, st, 1.0, float(iteration + 1))
        return gradient


# ------------------------------------------------------------------ regularization
class Regularization:
    JAVA_CLASS = None
    BEFORE_UPDATER, POST_UPDATER = "BEFORE_UPDATER", "POST_UPDATER"

    def __eq__(self, o):
        return type(self) is type(o) and self.__dict__ == o.__dict__

    def __hash__(self):
        return hash(type(self).__name__)


def _coef(x):
    return x if isinstance(x, ISchedule) else FixedSchedule(x)


class L1Regularization(Regularization):
    JAVA_CLASS = _REG + "L1Regularization"

    def __init__(self, l1):
        self.l1 = _coef(l1)

    def applyStep(self):
        return self.BEFORE_UPDATER

    def score(self, param, it=0, ep=0):
        return self.l1.valueAt(it, ep) * param.abs().sum()

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "l1": self.l1.to_json()}


class L2Regularization(Regularization):
    JAVA_CLASS = _REG + "L2Regularization"

    def __init__(self, l2):
        self.l2 = _coef(l2)

    def applyStep(self):
        return self.BEFORE_UPDATER

    def score(self, param, it=0, ep=0):
        return 0.5 * self.l2.valueAt(it, ep) * (param * param).sum()

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "l2": self.l2.to_json()}


class WeightDecay(Regularization):
    JAVA_CLASS = _REG + "WeightDecay"

    def __init__(self, coeff, applyLR=True):
        self.coeff, self.applyLR = _coef(coeff), bool(applyLR)

    def applyStep(self):
        return self.POST_UPDATER

    def score(self, param, it=0, ep=0):
        return 0.5 * self.coeff.valueAt(it, ep) * (param * param).sum()

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "coeff": self.coeff.to_json(), "applyLR": self.applyLR}


def regularization_from_json(lst):
    out = []
    for d in lst or []:
        c = d["@class"]
        if c.endswith("L1Regularization"):
            out.append(L1Regularization(schedule_from_json(d["l1"])))
        elif c.endswith("L2Regularization"):
            out.append(L2Regularization(schedule_from_json(d["l2"])))
        elif c.endswith("WeightDecay"):
            out.append(WeightDecay(schedule_from_json(d["coeff"]), d.get("applyLR", True)))
    return out
