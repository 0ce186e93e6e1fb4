"""Layer configuration classes (reference: deeplearning4j-nn/.../nn/conf/layers/**, 67 configs).

Each config knows: its Jackson "@class" name + JSON fields, output-shape inference
(``getOutputType``), nIn inference (``setNIn``), which InputPreProcessor it wants, and its flat
parameter layout (``param_specs`` — order & shapes are the checkpoint ABI, see
nn/params/*ParamInitializer.java and SURVEY Appendix B.3). ``instantiate`` creates the runtime layer.

Both construction styles work:  ``DenseLayer(nIn=4, nOut=3, activation="relu")`` and the fluent
``DenseLayer.Builder().nIn(4).nOut(3).activation(Activation.RELU).build()``.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional

from ...nd4j.activations import get_activation, IActivation, ActivationIdentity
from ...nd4j.lossfunctions import get_loss, ILossFunction
from ...nd4j.learning import (IUpdater, NoOp, updater_from_json, L1Regularization, L2Regularization, WeightDecay,
                              regularization_from_json, ISchedule)
from .base import (InputType, ConvolutionMode, PoolingType, GradientNormalization, get_weight_init, IWeightInit,
                   CNN2DFormat, RNNFormat, Distribution)

_L = "org.deeplearning4j.nn.conf.layers."


class ParamSpec:
    __slots__ = ("key", "shape", "order", "fan_in", "fan_out", "role", "init")

    def __init__(self, key, shape, order="c", fan_in=1, fan_out=1, role="W", init=None):
        self.key, self.shape, self.order = key, tuple(int(s) for s in shape), order
        self.fan_in, self.fan_out, self.role, self.init = fan_in, fan_out, role, init

    @property
    def numel(self):
        return int(math.prod(self.shape)) if self.shape else 1


def _pair(v, n=2):
    if v is None:
        return None
    if isinstance(v, (list, tuple)):
        if len(v) == 1:
            return tuple(int(v[0]) for _ in range(n))
        return tuple(int(x) for x in v)
    return tuple(int(v) for _ in range(n))


class _Builder:
    def __init__(self, cls, args, kw):
        object.__setattr__(self, "_cls", cls)
        object.__setattr__(self, "_kw", dict(kw))
        cls._builder_args(self._kw, *args)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)

        def setter(*v):
            self._kw[name] = v[0] if len(v) == 1 else list(v)
            return self
        return setter

    def build(self):
        return self._cls(**self._kw)


_ALIASES = {"name": "layerName", "activation": "activationFn", "weightInit": "weightInitFn", "updater": "iupdater",
            "dropOut": "idropout", "dropout": "idropout", "lossFunction": "lossFn", "lossFn": "lossFn",
            "nin": "nIn", "nout": "nOut", "dist": "dist", "eps": "eps", "poolingType": "poolingType",
            "units": "nOut", "cudnnAlgoMode": "cudnnAlgoMode", "dataFormat": "dataFormat"}

INHERITABLE = ("activationFn", "weightInitFn", "biasInit", "gainInit", "l1", "l2", "l1Bias", "l2Bias", "weightDecay",
               "weightDecayBias", "iupdater", "biasUpdater", "gradientNormalization", "gradientNormalizationThreshold",
               "idropout", "convolutionMode", "weightNoise", "constraints")


class Layer:
    """Base layer configuration."""
    JAVA_CLASS = _L + "Layer"
    FIELDS = dict(layerName=None, idropout=None, constraints=None)
    HAS_PARAMS = False

    def __init__(self, **kw):
        fields = {}
        for c in reversed(type(self).__mro__):
            fields.update(getattr(c, "FIELDS", {}))
        for k, v in fields.items():
            setattr(self, k, copy.copy(v))
        self._all_fields = list(fields.keys())
        dist = kw.pop("dist", None)
        for k, v in kw.items():
            k = _ALIASES.get(k, k)
            if k not in fields:
                raise TypeError("%s: unknown config field '%s'" % (type(self).__name__, k))
            setattr(self, k, v)
        if dist is not None:
            self.weightInitFn = get_weight_init(dist)
        self._normalize()

    # ---- construction helpers
    @classmethod
    def Builder(cls, *args, **kw):
        return _Builder(cls, args, kw)

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            raise TypeError("%s.Builder takes no positional arguments" % cls.__name__)

    def _normalize(self):
        if getattr(self, "activationFn", None) is not None:
            self.activationFn = get_activation(self.activationFn)
        if getattr(self, "weightInitFn", None) is not None:
            self.weightInitFn = get_weight_init(self.weightInitFn)
        d = getattr(self, "idropout", None)
        if d is not None and not isinstance(d, IDropout):
            self.idropout = Dropout(float(d)) if float(d) not in (0.0,) else None

    def clone(self):
        return copy.deepcopy(self)

    def getLayerName(self):
        return self.layerName

    # ---- global-default inheritance (NeuralNetConfiguration.Builder.configureLayer)
    def apply_global(self, g):
        for f in INHERITABLE:
            if hasattr(self, f) and getattr(self, f) is None and g.get(f) is not None:
                setattr(self, f, copy.deepcopy(g[f]))
        self._normalize()

    # ---- shapes
    def getOutputType(self, layerIndex, inputType):
        return inputType

    def setNIn(self, inputType, override):
        pass

    def getPreProcessorForInputType(self, inputType):
        return None

    def param_specs(self) -> List[ParamSpec]:
        return []

    def numParams(self):
        return sum(s.numel for s in self.param_specs())

    def isPretrainParam(self, key):
        return False

    # ---- regularization / updater lookups per parameter role
    def getRegularizationByParam(self, spec: ParamSpec):
        regs = []
        if spec.role in ("W", "gain"):
            l1, l2, wd = getattr(self, "l1", None), getattr(self, "l2", None), getattr(self, "weightDecay", None)
        elif spec.role == "b":
            l1, l2, wd = getattr(self, "l1Bias", None), getattr(self, "l2Bias", None), getattr(self, "weightDecayBias", None)
        else:
            return regs
        if l1:
            regs.append(L1Regularization(l1))
        if l2:
            regs.append(L2Regularization(l2))
        if wd:
            regs.append(wd if isinstance(wd, WeightDecay) else WeightDecay(wd, True))
        return regs

    def getUpdaterByParam(self, spec: ParamSpec) -> IUpdater:
        if spec.role == "stat":
            return NoOp()
        if spec.role == "b" and getattr(self, "biasUpdater", None) is not None:
            return self.biasUpdater
        return getattr(self, "iupdater", None) or NoOp()

    def instantiate(self, ctx):
        from ..layers import registry
        return registry.create(self, ctx)

    # ---- JSON
    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        for f in self._all_fields:
            v = getattr(self, f)
            d[_JSON_NAMES.get(f, f)] = _to_jsonable(v)
        regs = []
        regs_b = []
        for role, dst in (("W", regs), ("b", regs_b)):
            for r in self.getRegularizationByParam(ParamSpec("x", (1,), role=role)):
                dst.append(r.to_json())
        d["regularization"] = regs
        d["regularizationBias"] = regs_b
        for k in ("l1", "l2", "l1Bias", "l2Bias", "weightDecay", "weightDecayBias"):
            d.pop(k, None)
        return d

    @classmethod
    def from_json(cls, d):
        c = LAYER_CLASSES.get(d["@class"])
        if c is None:
            raise ValueError("Unknown layer class in JSON: %s" % d["@class"])
        kw = {}
        inv = {v: k for k, v in _JSON_NAMES.items()}
        fields = {}
        for k in reversed(c.__mro__):
            fields.update(getattr(k, "FIELDS", {}))
        for k, v in d.items():
            if k in ("@class", "regularization", "regularizationBias"):
                continue
            pk = inv.get(k, k)
            if pk not in fields:
                continue
            kw[pk] = _from_jsonable(pk, v)
        obj = c(**kw)
        for key, tgt in (("regularization", ("l1", "l2", "weightDecay")), ("regularizationBias", ("l1Bias", "l2Bias", "weightDecayBias"))):
            for r in regularization_from_json(d.get(key)):
                if isinstance(r, L1Regularization) and hasattr(obj, tgt[0]):
                    setattr(obj, tgt[0], r.l1.valueAt(0, 0))
                elif isinstance(r, L2Regularization) and hasattr(obj, tgt[1]):
                    setattr(obj, tgt[1], r.l2.valueAt(0, 0))
                elif isinstance(r, WeightDecay) and hasattr(obj, tgt[2]):
                    setattr(obj, tgt[2], r)
        return obj

    def __eq__(self, o):
        return type(self) is type(o) and self.to_json() == o.to_json()

    def __hash__(self):
        return hash(type(self).__name__)

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join(
            "%s=%r" % (f, getattr(self, f)) for f in self._all_fields if getattr(self, f) is not None))


_JSON_NAMES = {"nIn": "nin", "nOut": "nout"}


def _to_jsonable(v):
    if hasattr(v, "to_json"):
        return v.to_json()
    if isinstance(v, (list, tuple)):
        return [_to_jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: _to_jsonable(x) for k, x in v.items()}
    return v


def _from_jsonable(name, v):
    if isinstance(v, dict) and "@class" in v:
        c = v["@class"]
        if ".activations." in c:
            return get_activation(v)
        if ".lossfunctions." in c:
            return get_loss(v)
        if ".nn.weights." in c:
            return IWeightInit.from_json(v)
        if ".learning.config." in c:
            return updater_from_json(v)
        if ".conf.dropout." in c:
            return dropout_from_json(v)
        if ".conf.layers." in c or ".conf.layers.recurrent." in c or ".conf.layers.variational." in c:
            return Layer.from_json(v)
        if ".distribution." in c:
            from .base import distribution_from_json
            return distribution_from_json(v)
        if ".schedule." in c:
            from ...nd4j.learning import schedule_from_json
            return schedule_from_json(v)
    return v


# ------------------------------------------------------------------ dropout
_DP = "org.deeplearning4j.nn.conf.dropout."


class IDropout:
    JAVA_CLASS = None

    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        d.update({k: _to_jsonable(v) for k, v in self.__dict__.items() if not k.startswith("_")})
        return d

    def __eq__(self, o):
        return type(self) is type(o) and self.to_json() == o.to_json()

    def __hash__(self):
        return hash(type(self).__name__)


class Dropout(IDropout):
    """Inverted dropout; ``p`` is the RETAIN probability (DL4J convention)."""
    JAVA_CLASS = _DP + "Dropout"

    def __init__(self, p=0.5, pSchedule=None):
        self.p, self.pSchedule = float(p), pSchedule


class GaussianDropout(IDropout):
    JAVA_CLASS = _DP + "GaussianDropout"

    def __init__(self, rate=0.5, rateSchedule=None):
        self.rate, self.rateSchedule = float(rate), rateSchedule


class GaussianNoise(IDropout):
    JAVA_CLASS = _DP + "GaussianNoise"

    def __init__(self, stddev=0.1, stddevSchedule=None):
        self.stddev, self.stddevSchedule = float(stddev), stddevSchedule


class AlphaDropout(IDropout):
    JAVA_CLASS = _DP + "AlphaDropout"

    def __init__(self, p=0.5, pSchedule=None, alphaPrime=-1.7580993408473766, a=None, b=None):
        self.p, self.pSchedule, self.alphaPrime = float(p), pSchedule, alphaPrime


class SpatialDropout(IDropout):
    JAVA_CLASS = _DP + "SpatialDropout"

    def __init__(self, p=0.5, pSchedule=None):
        self.p, self.pSchedule = float(p), pSchedule


_DROPS = {c.JAVA_CLASS: c for c in (Dropout, GaussianDropout, GaussianNoise, AlphaDropout, SpatialDropout)}


def dropout_from_json(d):
    if d is None:
        return None
    cls = _DROPS[d["@class"]]
    kw = {k: v for k, v in d.items() if k != "@class" and k in cls.__init__.__code__.co_varnames}
    return cls(**kw)


# ------------------------------------------------------------------ base layers
class BaseLayer(Layer):
    JAVA_CLASS = _L + "BaseLayer"
    FIELDS = dict(activationFn=None, weightInitFn=None, biasInit=None, gainInit=None, l1=None, l2=None, l1Bias=None,
                  l2Bias=None, weightDecay=None, weightDecayBias=None, iupdater=None, biasUpdater=None,
                  gradientNormalization=None, gradientNormalizationThreshold=None, weightNoise=None)
    HAS_PARAMS = True


class FeedForwardLayer(BaseLayer):
    JAVA_CLASS = _L + "FeedForwardLayer"
    FIELDS = dict(nIn=0, nOut=0)

    def getOutputType(self, layerIndex, inputType):
        return InputType.feedForward(self.nOut)

    def setNIn(self, inputType, override):
        if self.nIn <= 0 or override:
            if inputType.kind in (InputType.FF, InputType.RNN):
                self.nIn = inputType.size
            else:
                self.nIn = inputType.arrayElementsPerExample()

    def getPreProcessorForInputType(self, inputType):
        from .preprocessors import CnnToFeedForwardPreProcessor, RnnToFeedForwardPreProcessor, Cnn3DToFeedForwardPreProcessor
        if inputType.kind == InputType.CNN:
            return CnnToFeedForwardPreProcessor(inputType.height, inputType.width, inputType.channels, inputType.format)
        if inputType.kind == InputType.CNN3D:
            return Cnn3DToFeedForwardPreProcessor(inputType.depth, inputType.height, inputType.width, inputType.channels)
        if inputType.kind == InputType.RNN:
            return RnnToFeedForwardPreProcessor()
        return None


class DenseLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "DenseLayer"
    FIELDS = dict(hasBias=True, hasLayerNorm=False)

    def param_specs(self):
        s = [ParamSpec("W", (self.nIn, self.nOut), "f", self.nIn, self.nOut, "W")]
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        if self.hasLayerNorm:
            s.append(ParamSpec("g", (1, self.nOut), "c", role="gain"))
        return s


class BaseOutputLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "BaseOutputLayer"
    FIELDS = dict(lossFn=None, hasBias=True)

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["lossFn"] = args[0]

    def _normalize(self):
        super()._normalize()
        if self.lossFn is not None:
            self.lossFn = get_loss(self.lossFn)
        elif type(self).__name__ != "BaseOutputLayer" and "lossFn" in self.FIELDS_ALL():
            self.lossFn = get_loss("MCXENT")

    @classmethod
    def FIELDS_ALL(cls):
        f = {}
        for c in reversed(cls.__mro__):
            f.update(getattr(c, "FIELDS", {}))
        return f

    def param_specs(self):
        s = [ParamSpec("W", (self.nIn, self.nOut), "f", self.nIn, self.nOut, "W")]
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        return s


class OutputLayer(BaseOutputLayer):
    JAVA_CLASS = _L + "OutputLayer"


class RnnOutputLayer(BaseOutputLayer):
    JAVA_CLASS = _L + "RnnOutputLayer"
    FIELDS = dict(rnnDataFormat=RNNFormat.NCW)

    def getOutputType(self, layerIndex, inputType):
        return InputType.recurrent(self.nOut, inputType.timeSeriesLength if inputType.kind == InputType.RNN else -1)

    def getPreProcessorForInputType(self, inputType):
        from .preprocessors import FeedForwardToRnnPreProcessor, CnnToRnnPreProcessor
        if inputType.kind == InputType.FF:
            return FeedForwardToRnnPreProcessor()
        if inputType.kind == InputType.CNN:
            return CnnToRnnPreProcessor(inputType.height, inputType.width, inputType.channels)
        return None


class CenterLossOutputLayer(BaseOutputLayer):
    JAVA_CLASS = _L + "CenterLossOutputLayer"
    FIELDS = dict(alpha=0.05, lambda_=2e-4, gradientCheck=False)

    def param_specs(self):
        return super().param_specs() + [ParamSpec("cL", (self.nOut, self.nIn), "c", role="stat",
                                                  init=lambda shape, gen, dt: __import__("torch").zeros(shape, dtype=dt))]

    def getUpdaterByParam(self, spec):
        if spec.key == "cL":
            return NoOp()
        return super().getUpdaterByParam(spec)


class LossLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "LossLayer"
    FIELDS = dict(lossFn=None)
    HAS_PARAMS = False

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["lossFn"] = args[0]

    def _normalize(self):
        super()._normalize()
        self.lossFn = get_loss(self.lossFn if self.lossFn is not None else "MCXENT")

    def getOutputType(self, layerIndex, inputType):
        return inputType

    def setNIn(self, inputType, override):
        pass

    def getPreProcessorForInputType(self, inputType):
        return None


class RnnLossLayer(LossLayer):
    JAVA_CLASS = _L + "RnnLossLayer"
    FIELDS = dict(rnnDataFormat=RNNFormat.NCW)


class CnnLossLayer(LossLayer):
    JAVA_CLASS = _L + "CnnLossLayer"
    FIELDS = dict(format=CNN2DFormat.NCHW)


class Cnn3DLossLayer(LossLayer):
    JAVA_CLASS = _L + "Cnn3DLossLayer"
    FIELDS = dict(dataFormat="NCDHW")


class ActivationLayer(Layer):
    JAVA_CLASS = _L + "ActivationLayer"
    FIELDS = dict(activationFn=None)

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["activationFn"] = args[0]


class DropoutLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "DropoutLayer"
    HAS_PARAMS = False

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["idropout"] = args[0]

    def getOutputType(self, layerIndex, inputType):
        return inputType

    def setNIn(self, inputType, override):
        pass

    def getPreProcessorForInputType(self, inputType):
        return None


class EmbeddingLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "EmbeddingLayer"
    FIELDS = dict(hasBias=False)

    def param_specs(self):
        s = [ParamSpec("W", (self.nIn, self.nOut), "f", self.nIn, self.nOut, "W")]
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        return s

    def setNIn(self, inputType, override):
        pass

    def getPreProcessorForInputType(self, inputType):
        return None


class EmbeddingSequenceLayer(EmbeddingLayer):
    JAVA_CLASS = _L + "EmbeddingSequenceLayer"
    FIELDS = dict(inputLength=1, inferInputLength=True, outputFormat=RNNFormat.NCW)

    def getOutputType(self, layerIndex, inputType):
        return InputType.recurrent(self.nOut, self.inputLength)


class ElementWiseMultiplicationLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "misc.ElementWiseMultiplicationLayer"

    def param_specs(self):
        return [ParamSpec("W", (1, self.nOut), "c", self.nIn, self.nOut, "W"), ParamSpec("b", (1, self.nOut), "c", role="b")]


class PReLULayer(BaseLayer):
    JAVA_CLASS = _L + "PReLULayer"
    FIELDS = dict(inputShape=None, sharedAxes=None)

    def param_specs(self):
        shape = list(self.inputShape)
        for a in (self.sharedAxes or []):
            shape[a - 1] = 1
        n = int(math.prod(self.inputShape))
        return [ParamSpec("W", tuple(shape), "c", n, n, "W")]

    def setNIn(self, inputType, override):
        if self.inputShape is None:
            self.inputShape = inputType.getShape(1)[1:]


class AutoEncoder(FeedForwardLayer):
    JAVA_CLASS = _L + "AutoEncoder"
    FIELDS = dict(corruptionLevel=0.3, sparsity=0.0, lossFn=None, visibleBiasInit=0.0)

    def _normalize(self):
        super()._normalize()
        self.lossFn = get_loss(self.lossFn if self.lossFn is not None else "MSE")

    def param_specs(self):
        return [ParamSpec("W", (self.nIn, self.nOut), "f", self.nIn, self.nOut, "W"),
                ParamSpec("b", (1, self.nOut), "c", role="b"), ParamSpec("vb", (1, self.nIn), "c", role="b")]

    def isPretrainParam(self, key):
        return key == "vb"


class VariationalAutoencoder(FeedForwardLayer):
    JAVA_CLASS = _L + "variational.VariationalAutoencoder"
    FIELDS = dict(encoderLayerSizes=(100,), decoderLayerSizes=(100,), reconstructionDistribution=None,
                  pzxActivationFn=None, numSamples=1)

    def _normalize(self):
        super()._normalize()
        self.pzxActivationFn = get_activation(self.pzxActivationFn)
        self.encoderLayerSizes = tuple(self.encoderLayerSizes)
        self.decoderLayerSizes = tuple(self.decoderLayerSizes)
        if self.reconstructionDistribution is None:
            self.reconstructionDistribution = {"type": "gaussian", "activation": "IDENTITY"}

    def _recon_size(self):
        t = self.reconstructionDistribution.get("type", "gaussian")
        return 2 * self.nIn if t == "gaussian" else self.nIn

    def param_specs(self):
        s = []
        prev = self.nIn
        for i, n in enumerate(self.encoderLayerSizes):
            s += [ParamSpec("e%dW" % i, (prev, n), "f", prev, n, "W"), ParamSpec("e%db" % i, (1, n), "c", role="b")]
            prev = n
        s += [ParamSpec("pZXMeanW", (prev, self.nOut), "f", prev, self.nOut, "W"), ParamSpec("pZXMeanb", (1, self.nOut), "c", role="b"),
              ParamSpec("pZXLogStd2W", (prev, self.nOut), "f", prev, self.nOut, "W"), ParamSpec("pZXLogStd2b", (1, self.nOut), "c", role="b")]
        prev = self.nOut
        for i, n in enumerate(self.decoderLayerSizes):
            s += [ParamSpec("d%dW" % i, (prev, n), "f", prev, n, "W"), ParamSpec("d%db" % i, (1, n), "c", role="b")]
            prev = n
        r = self._recon_size()
        s += [ParamSpec("pXZW", (prev, r), "f", prev, r, "W"), ParamSpec("pXZb", (1, r), "c", role="b")]
        return s

    def isPretrainParam(self, key):
        return key.startswith("d") or key.startswith("pXZ") or key.startswith("pZXLogStd2")


# ------------------------------------------------------------------ normalization
class BatchNormalization(FeedForwardLayer):
    JAVA_CLASS = _L + "BatchNormalization"
    FIELDS = dict(decay=0.9, eps=1e-5, isMinibatch=True, gamma=1.0, beta=0.0, lockGammaBeta=False,
                  cnn2DFormat=CNN2DFormat.NCHW, useLogStd=True)

    def getOutputType(self, layerIndex, inputType):
        return inputType

    def setNIn(self, inputType, override):
        if self.nIn <= 0 or override:
            n = inputType.channels if inputType.kind in (InputType.CNN, InputType.CNN3D, InputType.CNNFlat) else inputType.size
            self.nIn = self.nOut = n

    def getPreProcessorForInputType(self, inputType):
        from .preprocessors import FeedForwardToCnnPreProcessor
        if inputType.kind == InputType.CNNFlat:
            return FeedForwardToCnnPreProcessor(inputType.height, inputType.width, inputType.channels)
        return None

    def param_specs(self):
        import torch
        n = self.nOut
        s = []
        if not self.lockGammaBeta:
            s += [ParamSpec("gamma", (1, n), "c", role="W", init=lambda sh, g, dt: torch.full(sh, float(self.gamma), dtype=dt)),
                  ParamSpec("beta", (1, n), "c", role="W", init=lambda sh, g, dt: torch.full(sh, float(self.beta), dtype=dt))]
        s.append(ParamSpec("mean", (1, n), "c", role="stat", init=lambda sh, g, dt: torch.zeros(sh, dtype=dt)))
        if self.useLogStd:
            s.append(ParamSpec("log10stdev", (1, n), "c", role="stat", init=lambda sh, g, dt: torch.zeros(sh, dtype=dt)))
        else:
            s.append(ParamSpec("var", (1, n), "c", role="stat", init=lambda sh, g, dt: torch.ones(sh, dtype=dt)))
        return s

    def getRegularizationByParam(self, spec):
        return []  # no regularization on BN params (BatchNormalization.java:176-179)


class LocalResponseNormalization(Layer):
    JAVA_CLASS = _L + "LocalResponseNormalization"
    FIELDS = dict(n=5.0, k=2.0, beta=0.75, alpha=1e-4, dataFormat=CNN2DFormat.NCHW)


# ------------------------------------------------------------------ convolutional
def conv_out_size(i, k, s, p, d, mode):
    ke = k + (k - 1) * (d - 1)
    if mode == ConvolutionMode.Same:
        return int(math.ceil(i / float(s)))
    if mode == ConvolutionMode.Causal:
        return int(math.ceil(i / float(s)))
    if mode == ConvolutionMode.Strict and (i - ke + 2 * p) % s != 0:
        raise ValueError("ConvolutionMode.Strict: invalid size (in=%d,k=%d,s=%d,p=%d)" % (i, k, s, p))
    return (i - ke + 2 * p) // s + 1


def same_pad(i, k, s, d):
    ke = k + (k - 1) * (d - 1)
    o = int(math.ceil(i / float(s)))
    tot = max(0, (o - 1) * s + ke - i)
    return tot // 2, tot - tot // 2


class ConvolutionLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "ConvolutionLayer"
    FIELDS = dict(kernelSize=(5, 5), stride=(1, 1), padding=(0, 0), dilation=(1, 1), convolutionMode=None, hasBias=True,
                  cnn2dDataFormat=CNN2DFormat.NCHW, cudnnAlgoMode="PREFER_FASTEST", cudnnAllowFallback=True)
    NDIM = 2

    class AlgoMode:
        NO_WORKSPACE, PREFER_FASTEST, USER_SPECIFIED = "NO_WORKSPACE", "PREFER_FASTEST", "USER_SPECIFIED"

    @classmethod
    def _builder_args(cls, kw, *args):
        names = ("kernelSize", "stride", "padding")
        if len(args) > 3 or (args and not isinstance(args[0], (list, tuple))):
            kw["kernelSize"] = list(args)
            return
        for n, a in zip(names, args):
            kw[n] = a

    def _normalize(self):
        super()._normalize()
        n = self.NDIM
        self.kernelSize, self.stride = _pair(self.kernelSize, n), _pair(self.stride, n)
        self.padding, self.dilation = _pair(self.padding, n), _pair(self.dilation, n)

    def getOutputType(self, layerIndex, inputType):
        m = self.convolutionMode or ConvolutionMode.Truncate
        h = conv_out_size(inputType.height, self.kernelSize[0], self.stride[0], self.padding[0], self.dilation[0], m)
        w = conv_out_size(inputType.width, self.kernelSize[1], self.stride[1], self.padding[1], self.dilation[1], m)
        return InputType.convolutional(h, w, self.nOut, inputType.format)

    def setNIn(self, inputType, override):
        if self.nIn <= 0 or override:
            self.nIn = inputType.channels

    def getPreProcessorForInputType(self, inputType):
        from .preprocessors import FeedForwardToCnnPreProcessor, RnnToCnnPreProcessor
        if inputType.kind == InputType.CNNFlat:
            return FeedForwardToCnnPreProcessor(inputType.height, inputType.width, inputType.channels)
        return None

    def param_specs(self):
        k = self.kernelSize
        fi, fo = self.nIn * k[0] * k[1], self.nOut * k[0] * k[1] / (self.stride[0] * self.stride[1])
        s = []
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        s.append(ParamSpec("W", (self.nOut, self.nIn, k[0], k[1]), "c", fi, fo, "W"))
        return s


class Deconvolution2D(ConvolutionLayer):
    JAVA_CLASS = _L + "Deconvolution2D"

    def getOutputType(self, layerIndex, inputType):
        m = self.convolutionMode or ConvolutionMode.Truncate
        def o(i, k, s, p, d):
            if m == ConvolutionMode.Same:
                return i * s
            return s * (i - 1) + (k + (k - 1) * (d - 1)) - 2 * p
        return InputType.convolutional(o(inputType.height, self.kernelSize[0], self.stride[0], self.padding[0], self.dilation[0]),
                                       o(inputType.width, self.kernelSize[1], self.stride[1], self.padding[1], self.dilation[1]),
                                       self.nOut, inputType.format)

    def param_specs(self):
        k = self.kernelSize
        fi, fo = self.nIn * k[0] * k[1], self.nOut * k[0] * k[1]
        s = []
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        s.append(ParamSpec("W", (self.nIn, self.nOut, k[0], k[1]), "c", fi, fo, "W"))
        return s


class DepthwiseConvolution2D(ConvolutionLayer):
    JAVA_CLASS = _L + "DepthwiseConvolution2D"
    FIELDS = dict(depthMultiplier=1)

    def _normalize(self):
        super()._normalize()
        if self.nIn and self.nIn > 0:
            self.nOut = self.nIn * self.depthMultiplier

    def setNIn(self, inputType, override):
        super().setNIn(inputType, override)
        self.nOut = self.nIn * self.depthMultiplier

    def param_specs(self):
        k = self.kernelSize
        fi, fo = self.nIn * k[0] * k[1], self.depthMultiplier * k[0] * k[1]
        s = []
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        s.append(ParamSpec("W", (k[0], k[1], self.nIn, self.depthMultiplier), "c", fi, fo, "W"))
        return s


class SeparableConvolution2D(ConvolutionLayer):
    JAVA_CLASS = _L + "SeparableConvolution2D"
    FIELDS = dict(depthMultiplier=1)

    def param_specs(self):
        k = self.kernelSize
        fi, fo = self.nIn * k[0] * k[1], self.nOut * k[0] * k[1]
        s = []
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        s.append(ParamSpec("W", (self.depthMultiplier, self.nIn, k[0], k[1]), "c", fi, fo, "W"))
        s.append(ParamSpec("pW", (self.nOut, self.nIn * self.depthMultiplier, 1, 1), "c", fi, fo, "W"))
        return s


class Convolution1DLayer(ConvolutionLayer):
    JAVA_CLASS = _L + "Convolution1DLayer"
    FIELDS = dict(rnnDataFormat=RNNFormat.NCW)
    NDIM = 1

    def _normalize(self):
        BaseLayer._normalize(self)
        self.kernelSize, self.stride = _pair(self.kernelSize, 1)[:1], _pair(self.stride, 1)[:1]
        self.padding, self.dilation = _pair(self.padding, 1)[:1], _pair(self.dilation, 1)[:1]

    def getOutputType(self, layerIndex, inputType):
        m = self.convolutionMode or ConvolutionMode.Truncate
        t = inputType.timeSeriesLength
        o = conv_out_size(t, self.kernelSize[0], self.stride[0], self.padding[0], self.dilation[0], m) if t and t > 0 else -1
        return InputType.recurrent(self.nOut, o)

    def setNIn(self, inputType, override):
        if self.nIn <= 0 or override:
            self.nIn = inputType.size

    def getPreProcessorForInputType(self, inputType):
        return None

    def param_specs(self):
        k = self.kernelSize[0]
        s = []
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        s.append(ParamSpec("W", (self.nOut, self.nIn, k, 1), "c", self.nIn * k, self.nOut * k, "W"))
        return s


Convolution1D = Convolution1DLayer
Convolution2D = ConvolutionLayer


class Convolution3D(ConvolutionLayer):
    JAVA_CLASS = _L + "Convolution3D"
    FIELDS = dict(kernelSize=(2, 2, 2), stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), dataFormat="NCDHW")
    NDIM = 3

    def getOutputType(self, layerIndex, inputType):
        m = self.convolutionMode or ConvolutionMode.Truncate
        dims = [conv_out_size(i, self.kernelSize[j], self.stride[j], self.padding[j], self.dilation[j], m)
                for j, i in enumerate((inputType.depth, inputType.height, inputType.width))]
        return InputType.convolutional3D(dims[0], dims[1], dims[2], self.nOut)

    def param_specs(self):
        k = self.kernelSize
        kk = k[0] * k[1] * k[2]
        s = []
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        s.append(ParamSpec("W", (self.nOut, self.nIn, k[0], k[1], k[2]), "c", self.nIn * kk, self.nOut * kk, "W"))
        return s


class Deconvolution3D(Convolution3D):
    JAVA_CLASS = _L + "Deconvolution3D"

    def getOutputType(self, layerIndex, inputType):
        m = self.convolutionMode or ConvolutionMode.Truncate
        def o(i, j):
            if m == ConvolutionMode.Same:
                return i * self.stride[j]
            return self.stride[j] * (i - 1) + self.kernelSize[j] - 2 * self.padding[j]
        return InputType.convolutional3D(o(inputType.depth, 0), o(inputType.height, 1), o(inputType.width, 2), self.nOut)

    def param_specs(self):
        k = self.kernelSize
        kk = k[0] * k[1] * k[2]
        s = []
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        s.append(ParamSpec("W", (self.nIn, self.nOut, k[0], k[1], k[2]), "c", self.nIn * kk, self.nOut * kk, "W"))
        return s


class SubsamplingLayer(Layer):
    JAVA_CLASS = _L + "SubsamplingLayer"
    FIELDS = dict(poolingType=PoolingType.MAX, kernelSize=(1, 1), stride=(2, 2), padding=(0, 0), dilation=(1, 1),
                  convolutionMode=None, pnorm=0, eps=1e-8, cnn2dDataFormat=CNN2DFormat.NCHW, avgPoolIncludePadInDivisor=True)
    NDIM = 2

    @classmethod
    def _builder_args(cls, kw, *args):
        args = list(args)
        if args and isinstance(args[0], str):
            kw["poolingType"] = args.pop(0)
        for n, a in zip(("kernelSize", "stride", "padding"), args):
            kw[n] = a

    def _normalize(self):
        super()._normalize()
        n = self.NDIM
        self.poolingType = str(self.poolingType).upper()
        self.kernelSize, self.stride = _pair(self.kernelSize, n), _pair(self.stride, n)
        self.padding, self.dilation = _pair(self.padding, n), _pair(self.dilation, n)

    def getOutputType(self, layerIndex, inputType):
        m = self.convolutionMode or ConvolutionMode.Truncate
        h = conv_out_size(inputType.height, self.kernelSize[0], self.stride[0], self.padding[0], self.dilation[0], m)
        w = conv_out_size(inputType.width, self.kernelSize[1], self.stride[1], self.padding[1], self.dilation[1], m)
        return InputType.convolutional(h, w, inputType.channels, inputType.format)

    def getPreProcessorForInputType(self, inputType):
        from .preprocessors import FeedForwardToCnnPreProcessor
        if inputType.kind == InputType.CNNFlat:
            return FeedForwardToCnnPreProcessor(inputType.height, inputType.width, inputType.channels)
        return None


class Subsampling1DLayer(SubsamplingLayer):
    JAVA_CLASS = _L + "Subsampling1DLayer"
    NDIM = 1

    def _normalize(self):
        Layer._normalize(self)
        self.poolingType = str(self.poolingType).upper()
        self.kernelSize, self.stride = _pair(self.kernelSize, 1)[:1], _pair(self.stride, 1)[:1]
        self.padding, self.dilation = _pair(self.padding, 1)[:1], _pair(self.dilation, 1)[:1]

    def getOutputType(self, layerIndex, inputType):
        m = self.convolutionMode or ConvolutionMode.Truncate
        t = inputType.timeSeriesLength
        o = conv_out_size(t, self.kernelSize[0], self.stride[0], self.padding[0], self.dilation[0], m) if t and t > 0 else -1
        return InputType.recurrent(inputType.size, o)

    def getPreProcessorForInputType(self, inputType):
        return None


class Subsampling3DLayer(SubsamplingLayer):
    JAVA_CLASS = _L + "Subsampling3DLayer"
    FIELDS = dict(kernelSize=(1, 1, 1), stride=(2, 2, 2), padding=(0, 0, 0), dilation=(1, 1, 1), dataFormat="NCDHW")
    NDIM = 3

    def getOutputType(self, layerIndex, inputType):
        m = self.convolutionMode or ConvolutionMode.Truncate
        dims = [conv_out_size(i, self.kernelSize[j], self.stride[j], self.padding[j], self.dilation[j], m)
                for j, i in enumerate((inputType.depth, inputType.height, inputType.width))]
        return InputType.convolutional3D(dims[0], dims[1], dims[2], inputType.channels)

    def getPreProcessorForInputType(self, inputType):
        return None


class Upsampling2D(Layer):
    JAVA_CLASS = _L + "Upsampling2D"
    FIELDS = dict(size=(2, 2), format=CNN2DFormat.NCHW)

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["size"] = args[0] if len(args) == 1 else list(args)

    def _normalize(self):
        super()._normalize()
        self.size = _pair(self.size, 2)

    def getOutputType(self, layerIndex, inputType):
        return InputType.convolutional(inputType.height * self.size[0], inputType.width * self.size[1], inputType.channels, inputType.format)


class Upsampling1D(Layer):
    JAVA_CLASS = _L + "Upsampling1D"
    FIELDS = dict(size=(2,))

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["size"] = args[0]

    def _normalize(self):
        super()._normalize()
        self.size = _pair(self.size, 1)[:1]

    def getOutputType(self, layerIndex, inputType):
        t = inputType.timeSeriesLength
        return InputType.recurrent(inputType.size, t * self.size[0] if t and t > 0 else -1)


class Upsampling3D(Layer):
    JAVA_CLASS = _L + "Upsampling3D"
    FIELDS = dict(size=(2, 2, 2), dataFormat="NCDHW")

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["size"] = args[0] if len(args) == 1 else list(args)

    def _normalize(self):
        super()._normalize()
        self.size = _pair(self.size, 3)

    def getOutputType(self, layerIndex, inputType):
        return InputType.convolutional3D(inputType.depth * self.size[0], inputType.height * self.size[1],
                                         inputType.width * self.size[2], inputType.channels)


def _pad4(p):
    p = list(p) if isinstance(p, (list, tuple)) else [p]
    if len(p) == 1:
        return (p[0],) * 4
    if len(p) == 2:
        return (p[0], p[0], p[1], p[1])
    return tuple(int(x) for x in p)


class ZeroPaddingLayer(Layer):
    JAVA_CLASS = _L + "ZeroPaddingLayer"
    FIELDS = dict(padding=(0, 0, 0, 0), dataFormat=CNN2DFormat.NCHW)  # [top, bottom, left, right]

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["padding"] = args[0] if len(args) == 1 and isinstance(args[0], (list, tuple)) else list(args)

    def _normalize(self):
        super()._normalize()
        self.padding = _pad4(self.padding)

    def getOutputType(self, layerIndex, inputType):
        p = self.padding
        return InputType.convolutional(inputType.height + p[0] + p[1], inputType.width + p[2] + p[3], inputType.channels, inputType.format)


class ZeroPadding1DLayer(Layer):
    JAVA_CLASS = _L + "ZeroPadding1DLayer"
    FIELDS = dict(padding=(0, 0))

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["padding"] = list(args)

    def _normalize(self):
        super()._normalize()
        self.padding = _pair(self.padding, 2)

    def getOutputType(self, layerIndex, inputType):
        t = inputType.timeSeriesLength
        return InputType.recurrent(inputType.size, t + sum(self.padding) if t and t > 0 else -1)


class ZeroPadding3DLayer(Layer):
    JAVA_CLASS = _L + "ZeroPadding3DLayer"
    FIELDS = dict(padding=(0, 0, 0, 0, 0, 0))

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["padding"] = list(args)

    def _normalize(self):
        super()._normalize()
        p = list(self.padding)
        if len(p) == 3:
            p = [p[0], p[0], p[1], p[1], p[2], p[2]]
        self.padding = tuple(p)

    def getOutputType(self, layerIndex, inputType):
        p = self.padding
        return InputType.convolutional3D(inputType.depth + p[0] + p[1], inputType.height + p[2] + p[3],
                                         inputType.width + p[4] + p[5], inputType.channels)


class Cropping2D(Layer):
    JAVA_CLASS = _L + "convolutional.Cropping2D"
    FIELDS = dict(cropping=(0, 0, 0, 0), dataFormat=CNN2DFormat.NCHW)

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["cropping"] = args[0] if len(args) == 1 and isinstance(args[0], (list, tuple)) else list(args)

    def _normalize(self):
        super()._normalize()
        self.cropping = _pad4(self.cropping)

    def getOutputType(self, layerIndex, inputType):
        c = self.cropping
        return InputType.convolutional(inputType.height - c[0] - c[1], inputType.width - c[2] - c[3], inputType.channels, inputType.format)


class Cropping1D(Layer):
    JAVA_CLASS = _L + "convolutional.Cropping1D"
    FIELDS = dict(cropping=(0, 0))

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["cropping"] = list(args)

    def _normalize(self):
        super()._normalize()
        self.cropping = _pair(self.cropping, 2)

    def getOutputType(self, layerIndex, inputType):
        t = inputType.timeSeriesLength
        return InputType.recurrent(inputType.size, t - sum(self.cropping) if t and t > 0 else -1)


class Cropping3D(Layer):
    JAVA_CLASS = _L + "convolutional.Cropping3D"
    FIELDS = dict(cropping=(0, 0, 0, 0, 0, 0))

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["cropping"] = list(args)

    def _normalize(self):
        super()._normalize()
        p = list(self.cropping)
        if len(p) == 3:
            p = [p[0], p[0], p[1], p[1], p[2], p[2]]
        self.cropping = tuple(p)

    def getOutputType(self, layerIndex, inputType):
        c = self.cropping
        return InputType.convolutional3D(inputType.depth - c[0] - c[1], inputType.height - c[2] - c[3],
                                         inputType.width - c[4] - c[5], inputType.channels)


class SpaceToDepthLayer(Layer):
    JAVA_CLASS = _L + "SpaceToDepthLayer"
    FIELDS = dict(blockSize=2, dataFormat=CNN2DFormat.NCHW)

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["blockSize"] = args[0]

    def getOutputType(self, layerIndex, inputType):
        b = self.blockSize
        return InputType.convolutional(inputType.height // b, inputType.width // b, inputType.channels * b * b, inputType.format)


class SpaceToBatchLayer(Layer):
    JAVA_CLASS = _L + "SpaceToBatchLayer"
    FIELDS = dict(blocks=(2, 2), padding=((0, 0), (0, 0)), format=CNN2DFormat.NCHW)

    def getOutputType(self, layerIndex, inputType):
        b, p = self.blocks, self.padding
        return InputType.convolutional((inputType.height + p[0][0] + p[0][1]) // b[0],
                                       (inputType.width + p[1][0] + p[1][1]) // b[1], inputType.channels, inputType.format)


class GlobalPoolingLayer(Layer):
    JAVA_CLASS = _L + "GlobalPoolingLayer"
    FIELDS = dict(poolingType=PoolingType.MAX, poolingDimensions=None, pnorm=2, collapseDimensions=True)

    @classmethod
    def _builder_args(cls, kw, *args):
        if args:
            kw["poolingType"] = args[0]

    def _normalize(self):
        super()._normalize()
        self.poolingType = str(self.poolingType).upper()

    def getOutputType(self, layerIndex, inputType):
        if inputType.kind == InputType.RNN:
            return InputType.feedForward(inputType.size) if self.collapseDimensions else InputType.recurrent(inputType.size, 1)
        if inputType.kind == InputType.CNN:
            return InputType.feedForward(inputType.channels) if self.collapseDimensions else InputType.convolutional(1, 1, inputType.channels)
        if inputType.kind == InputType.CNN3D:
            return InputType.feedForward(inputType.channels) if self.collapseDimensions else InputType.convolutional3D(1, 1, 1, inputType.channels)
        return inputType


class LocallyConnected2D(ConvolutionLayer):
    JAVA_CLASS = _L + "LocallyConnected2D"
    FIELDS = dict(inputSize=None, outputSize=None)

    def setNIn(self, inputType, override):
        super().setNIn(inputType, override)
        self.inputSize = (inputType.height, inputType.width)
        o = self.getOutputType(0, inputType)
        self.outputSize = (o.height, o.width)

    def param_specs(self):
        k = self.kernelSize
        n_out_pos = self.outputSize[0] * self.outputSize[1]
        fd = self.nIn * k[0] * k[1]
        s = [ParamSpec("W", (n_out_pos, fd, self.nOut), "c", fd, self.nOut, "W")]
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        return s


class LocallyConnected1D(Convolution1DLayer):
    JAVA_CLASS = _L + "LocallyConnected1D"
    FIELDS = dict(inputSize=None, outputSize=None)

    def setNIn(self, inputType, override):
        super().setNIn(inputType, override)
        self.inputSize = inputType.timeSeriesLength
        self.outputSize = self.getOutputType(0, inputType).timeSeriesLength

    def param_specs(self):
        fd = self.nIn * self.kernelSize[0]
        s = [ParamSpec("W", (self.outputSize, fd, self.nOut), "c", fd, self.nOut, "W")]
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        return s


# ------------------------------------------------------------------ recurrent
class BaseRecurrentLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "BaseRecurrentLayer"
    FIELDS = dict(weightInitFnRecurrent=None, rnnDataFormat=RNNFormat.NCW)

    def _normalize(self):
        super()._normalize()
        if self.weightInitFnRecurrent is not None:
            self.weightInitFnRecurrent = get_weight_init(self.weightInitFnRecurrent)

    def getOutputType(self, layerIndex, inputType):
        return InputType.recurrent(self.nOut, inputType.timeSeriesLength if inputType.kind == InputType.RNN else -1)

    def setNIn(self, inputType, override):
        if self.nIn <= 0 or override:
            self.nIn = inputType.size if inputType.kind in (InputType.RNN, InputType.FF) else inputType.arrayElementsPerExample()

    def getPreProcessorForInputType(self, inputType):
        from .preprocessors import FeedForwardToRnnPreProcessor, CnnToRnnPreProcessor
        if inputType.kind == InputType.FF:
            return FeedForwardToRnnPreProcessor()
        if inputType.kind == InputType.CNN:
            return CnnToRnnPreProcessor(inputType.height, inputType.width, inputType.channels)
        return None


class LSTM(BaseRecurrentLayer):
    JAVA_CLASS = _L + "LSTM"
    FIELDS = dict(forgetGateBiasInit=1.0, gateActivationFn=None)
    PEEPHOLE = 0

    def _normalize(self):
        super()._normalize()
        self.gateActivationFn = get_activation(self.gateActivationFn if self.gateActivationFn is not None else "SIGMOID")

    def param_specs(self):
        import torch
        n, nin = self.nOut, self.nIn
        fgb = float(self.forgetGateBiasInit)

        def binit(sh, g, dt):
            b = torch.zeros(sh, dtype=dt)
            b[0, n:2 * n] = fgb
            return b
        return [ParamSpec("W", (nin, 4 * n), "f", nin, n, "W"),
                ParamSpec("RW", (n, 4 * n + self.PEEPHOLE), "f", n, n, "RW"),
                ParamSpec("b", (1, 4 * n), "c", role="b", init=binit)]

    def getRegularizationByParam(self, spec):
        if spec.role == "RW":
            return super().getRegularizationByParam(ParamSpec("x", (1,), role="W"))
        return super().getRegularizationByParam(spec)


class GravesLSTM(LSTM):
    JAVA_CLASS = _L + "GravesLSTM"
    PEEPHOLE = 3


class GravesBidirectionalLSTM(LSTM):
    JAVA_CLASS = _L + "GravesBidirectionalLSTM"
    PEEPHOLE = 3

    def param_specs(self):
        out = []
        for d in ("F", "B"):
            for s in super().param_specs():
                s.key = {"W": "W" + d, "RW": "RW" + d, "b": "b" + d}[s.key]
                out.append(s)
        return out


class SimpleRnn(BaseRecurrentLayer):
    JAVA_CLASS = _L + "recurrent.SimpleRnn"
    FIELDS = dict(hasLayerNorm=False)

    def param_specs(self):
        n, nin = self.nOut, self.nIn
        s = [ParamSpec("W", (nin, n), "f", nin, n, "W"), ParamSpec("RW", (n, n), "f", n, n, "RW"),
             ParamSpec("b", (1, n), "c", role="b")]
        if self.hasLayerNorm:
            s.append(ParamSpec("g", (1, 2 * n), "c", role="gain"))
        return s

    def getRegularizationByParam(self, spec):
        if spec.role == "RW":
            return super().getRegularizationByParam(ParamSpec("x", (1,), role="W"))
        return super().getRegularizationByParam(spec)


class _WrapperLayer(Layer):
    FIELDS = dict(underlying=None)
    HAS_PARAMS = True

    def _normalize(self):
        super()._normalize()
        if isinstance(self.underlying, dict):
            self.underlying = Layer.from_json(self.underlying)

    def apply_global(self, g):
        super().apply_global(g)
        if self.underlying is not None:
            self.underlying.apply_global(g)

    def getOutputType(self, layerIndex, inputType):
        return self.underlying.getOutputType(layerIndex, inputType)

    def setNIn(self, inputType, override):
        self.underlying.setNIn(inputType, override)

    def getPreProcessorForInputType(self, inputType):
        return self.underlying.getPreProcessorForInputType(inputType)

    def param_specs(self):
        return self.underlying.param_specs()

    def getRegularizationByParam(self, spec):
        return self.underlying.getRegularizationByParam(spec)

    def getUpdaterByParam(self, spec):
        return self.underlying.getUpdaterByParam(spec)


class Bidirectional(_WrapperLayer):
    JAVA_CLASS = _L + "recurrent.Bidirectional"
    FIELDS = dict(mode="CONCAT", fwd=None, bwd=None)

    class Mode:
        ADD, MUL, AVERAGE, CONCAT = "ADD", "MUL", "AVERAGE", "CONCAT"

    def __init__(self, *args, **kw):
        if args:
            if len(args) == 2:
                kw["mode"], kw["fwd"] = args
            else:
                kw["fwd"] = args[0]
        super().__init__(**kw)

    def _normalize(self):
        Layer._normalize(self)
        if isinstance(self.fwd, dict):
            self.fwd = Layer.from_json(self.fwd)
        if isinstance(self.bwd, dict):
            self.bwd = Layer.from_json(self.bwd)
        if self.fwd is not None and self.bwd is None:
            self.bwd = self.fwd.clone()
        self.underlying = self.fwd

    def apply_global(self, g):
        Layer.apply_global(self, g)
        self.fwd.apply_global(g)
        self.bwd.apply_global(g)

    def setNIn(self, inputType, override):
        self.fwd.setNIn(inputType, override)
        self.bwd.setNIn(inputType, override)

    def getOutputType(self, layerIndex, inputType):
        o = self.fwd.getOutputType(layerIndex, inputType)
        if self.mode == "CONCAT":
            return InputType.recurrent(2 * o.size, o.timeSeriesLength)
        return o

    def param_specs(self):
        out = []
        for pre, l in (("f", self.fwd), ("b", self.bwd)):
            for s in l.param_specs():
                s.key = pre + s.key
                out.append(s)
        return out

    def _sub_for(self, spec):
        return self.fwd if spec.key.startswith("f") else self.bwd

    def getRegularizationByParam(self, spec):
        return self.fwd.getRegularizationByParam(spec)

    def getUpdaterByParam(self, spec):
        return self.fwd.getUpdaterByParam(spec)


class LastTimeStep(_WrapperLayer):
    JAVA_CLASS = _L + "recurrent.LastTimeStep"

    def __init__(self, underlying=None, **kw):
        super().__init__(underlying=underlying, **kw)

    def getOutputType(self, layerIndex, inputType):
        o = self.underlying.getOutputType(layerIndex, inputType)
        return InputType.feedForward(o.size)


class TimeDistributed(_WrapperLayer):
    JAVA_CLASS = _L + "recurrent.TimeDistributed"
    FIELDS = dict(rnnDataFormat=RNNFormat.NCW)

    def __init__(self, underlying=None, **kw):
        super().__init__(underlying=underlying, **kw)

    def getOutputType(self, layerIndex, inputType):
        ff = InputType.feedForward(inputType.size)
        o = self.underlying.getOutputType(layerIndex, ff)
        return InputType.recurrent(o.size, inputType.timeSeriesLength)

    def setNIn(self, inputType, override):
        self.underlying.setNIn(InputType.feedForward(inputType.size), override)

    def getPreProcessorForInputType(self, inputType):
        return None


class MaskZeroLayer(_WrapperLayer):
    JAVA_CLASS = _L + "util.MaskZeroLayer"
    FIELDS = dict(maskingValue=0.0)

    def __init__(self, underlying=None, **kw):
        super().__init__(underlying=underlying, **kw)


class FrozenLayer(_WrapperLayer):
    JAVA_CLASS = _L + "misc.FrozenLayer"
    FIELDS = dict(layer=None)

    def __init__(self, layer=None, **kw):
        kw.setdefault("underlying", layer)
        super().__init__(layer=layer, **kw)

    def _normalize(self):
        Layer._normalize(self)
        if isinstance(self.layer, dict):
            self.layer = Layer.from_json(self.layer)
        self.underlying = self.layer

    def getUpdaterByParam(self, spec):
        return NoOp()

    def getRegularizationByParam(self, spec):
        return []


class FrozenLayerWithBackprop(FrozenLayer):
    JAVA_CLASS = _L + "misc.FrozenLayerWithBackprop"


class MaskLayer(Layer):
    JAVA_CLASS = _L + "util.MaskLayer"


class RepeatVector(FeedForwardLayer):
    JAVA_CLASS = _L + "misc.RepeatVector"
    FIELDS = dict(n=1, dataFormat=RNNFormat.NCW)
    HAS_PARAMS = False

    def getOutputType(self, layerIndex, inputType):
        return InputType.recurrent(inputType.size, self.n)

    def setNIn(self, inputType, override):
        pass

    def getPreProcessorForInputType(self, inputType):
        return None


# ------------------------------------------------------------------ attention (SameDiff layers in the reference)
class SelfAttentionLayer(FeedForwardLayer):
    JAVA_CLASS = _L + "SelfAttentionLayer"
    FIELDS = dict(nHeads=1, headSize=0, projectInput=True)

    def _hs(self):
        return self.headSize if self.headSize else self.nOut // self.nHeads

    def getOutputType(self, layerIndex, inputType):
        return InputType.recurrent(self.nOut if self.projectInput else inputType.size, inputType.timeSeriesLength)

    def setNIn(self, inputType, override):
        if self.nIn <= 0 or override:
            self.nIn = inputType.size
        if not self.projectInput:
            self.nOut = self.nIn

    def getPreProcessorForInputType(self, inputType):
        return None

    def param_specs(self):
        if not self.projectInput:
            return []
        h, hs = self.nHeads, self._hs()
        return [ParamSpec("Wq", (h, hs, self.nIn), "c", self.nIn, hs, "W"), ParamSpec("Wk", (h, hs, self.nIn), "c", self.nIn, hs, "W"),
                ParamSpec("Wv", (h, hs, self.nIn), "c", self.nIn, hs, "W"), ParamSpec("Wo", (h * hs, self.nOut), "c", h * hs, self.nOut, "W")]


class LearnedSelfAttentionLayer(SelfAttentionLayer):
    JAVA_CLASS = _L + "LearnedSelfAttentionLayer"
    FIELDS = dict(nQueries=1)

    def getOutputType(self, layerIndex, inputType):
        return InputType.recurrent(self.nOut if self.projectInput else inputType.size, self.nQueries)

    def param_specs(self):
        s = super().param_specs()
        s.append(ParamSpec("Q", (1, self.nIn, self.nQueries), "c", self.nIn, self.nQueries, "W"))
        return s


class RecurrentAttentionLayer(SelfAttentionLayer):
    JAVA_CLASS = _L + "RecurrentAttentionLayer"
    FIELDS = dict(hasBias=True)

    def getOutputType(self, layerIndex, inputType):
        return InputType.recurrent(self.nOut, inputType.timeSeriesLength)

    def param_specs(self):
        h, hs = self.nHeads, self._hs()
        s = [ParamSpec("W", (self.nIn, self.nOut), "c", self.nIn, self.nOut, "W"),
             ParamSpec("R", (self.nOut, self.nOut), "c", self.nOut, self.nOut, "W")]
        if self.hasBias:
            s.append(ParamSpec("b", (1, self.nOut), "c", role="b"))
        if self.projectInput:
            s += [ParamSpec("Wq", (h, hs, self.nOut), "c", self.nOut, hs, "W"), ParamSpec("Wk", (h, hs, self.nIn), "c", self.nIn, hs, "W"),
                  ParamSpec("Wv", (h, hs, self.nIn), "c", self.nIn, hs, "W"), ParamSpec("Wo", (h * hs, self.nOut), "c", h * hs, self.nOut, "W")]
        return s


class DotProductAttentionLayer(Layer):
    JAVA_CLASS = _L + "DotProductAttentionLayer"
    FIELDS = dict(scale=0.0, dropoutProbability=0.0, useCausalMask=False, nIn=0, nOut=0)


# ------------------------------------------------------------------ capsules / yolo / misc
class PrimaryCapsules(ConvolutionLayer):
    JAVA_CLASS = _L + "PrimaryCapsules"
    FIELDS = dict(capsuleDimensions=8, channels=32, useRelu=False, leak=0.0, capsules=0)

    def _normalize(self):
        super()._normalize()
        self.nOut = self.capsuleDimensions * self.channels

    def getOutputType(self, layerIndex, inputType):
        o = ConvolutionLayer.getOutputType(self, layerIndex, inputType)
        self.capsules = o.height * o.width * self.channels
        return InputType.recurrent(self.capsules, self.capsuleDimensions)


class CapsuleLayer(BaseLayer):
    JAVA_CLASS = _L + "CapsuleLayer"
    FIELDS = dict(capsules=10, capsuleDimensions=16, routings=3, inputCapsules=0, inputCapsuleDimensions=0, hasBias=False)

    def setNIn(self, inputType, override):
        self.inputCapsules, self.inputCapsuleDimensions = inputType.size, inputType.timeSeriesLength

    def getOutputType(self, layerIndex, inputType):
        return InputType.recurrent(self.capsules, self.capsuleDimensions)

    def param_specs(self):
        s = [ParamSpec("W", (1, self.inputCapsules, self.capsules * self.capsuleDimensions, self.inputCapsuleDimensions, 1), "c",
                       self.inputCapsuleDimensions, self.capsuleDimensions, "W")]
        if self.hasBias:
            s.append(ParamSpec("b", (1, 1, self.capsules, self.capsuleDimensions, 1), "c", role="b"))
        return s


class CapsuleStrengthLayer(Layer):
    JAVA_CLASS = _L + "CapsuleStrengthLayer"

    def getOutputType(self, layerIndex, inputType):
        return InputType.feedForward(inputType.size)


class Yolo2OutputLayer(Layer):
    JAVA_CLASS = _L + "objdetect.Yolo2OutputLayer"
    FIELDS = dict(lambdaCoord=5.0, lambdaNoObj=0.5, boundingBoxes=None, lossPositionScale=None, lossClassPredictions=None)

    def _normalize(self):
        super()._normalize()


class OCNNOutputLayer(BaseOutputLayer):
    JAVA_CLASS = "org.deeplearning4j.nn.conf.ocnn.OCNNOutputLayer"
    FIELDS = dict(hiddenSize=10, nu=0.04, windowSize=10000, initialRValue=0.1, configureR=True)

    def param_specs(self):
        return [ParamSpec("W", (self.nIn, self.hiddenSize), "f", self.nIn, self.hiddenSize, "W"),
                ParamSpec("V", (self.hiddenSize, 1), "f", self.hiddenSize, 1, "W"),
                ParamSpec("R", (1, 1), "c", role="stat", init=lambda sh, g, dt: __import__("torch").full(sh, float(self.initialRValue), dtype=dt))]

    def getOutputType(self, layerIndex, inputType):
        return InputType.feedForward(1)


class SameDiffLambdaLayer(Layer):
    """LambdaLayer: parameter-free layer defined by a Python callable on torch tensors
    (reference: nn/conf/layers/samediff/SameDiffLambdaLayer.java)."""
    JAVA_CLASS = _L + "samediff.SameDiffLambdaLayer"
    FIELDS = dict(fn=None, outputTypeFn=None)

    def getOutputType(self, layerIndex, inputType):
        return self.outputTypeFn(inputType) if self.outputTypeFn else inputType

    def to_json(self):
        d = super().to_json()
        d.pop("fn", None)
        d.pop("outputTypeFn", None)
        return d


class SameDiffLayer(BaseLayer):
    """User-defined layer with parameters: subclass and implement defineParameters() -> {name: shape},
    defineLayer(x, params, mask) (reference: nn/conf/layers/samediff/SameDiffLayer.java)."""
    JAVA_CLASS = _L + "samediff.SameDiffLayer"
    FIELDS = dict(nIn=0, nOut=0)

    def defineParameters(self):
        return {}

    def defineLayer(self, x, params, mask=None):
        raise NotImplementedError

    def param_specs(self):
        return [ParamSpec(k, shp, "c", shp[0] if shp else 1, shp[-1] if shp else 1, "b" if k.startswith("b") else "W")
                for k, shp in self.defineParameters().items()]


class SameDiffOutputLayer(SameDiffLayer):
    """User-defined output layer: implement defineParameters() and
    ``defineLayer(x, labels, params) -> (activations, per_example_loss[mb])`` with torch ops; the backward pass comes from
    autograd (reference: nn/conf/layers/samediff/SameDiffOutputLayer.java — defineLayer returns the loss variable,
    activationsVertexName names the prediction)."""
    JAVA_CLASS = _L + "samediff.SameDiffOutputLayer"

    def defineLayer(self, x, labels, params):
        raise NotImplementedError

    def labelsRequired(self):
        return True


LambdaLayer = SameDiffLambdaLayer


def _collect():
    out = {}
    stack = [Layer]
    while stack:
        c = stack.pop()
        out[c.JAVA_CLASS] = c
        stack.extend(c.__subclasses__())
    return out


LAYER_CLASSES = _collect()


def register_layer_class(cls):
    LAYER_CLASSES[cls.JAVA_CLASS] = cls
    return cls
