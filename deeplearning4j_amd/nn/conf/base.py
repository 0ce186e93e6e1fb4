"""Configuration primitives: enums, InputType, distributions, weight initialisation.

Reference: deeplearning4j-nn/.../nn/conf/inputs/InputType.java, nn/conf/distribution/*,
nn/weights/WeightInit*.java (formulas), nn/conf/{ConvolutionMode,GradientNormalization,...}.java.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch


class ConvolutionMode:
    Strict, Truncate, Same, Causal = "Strict", "Truncate", "Same", "Causal"


class PoolingType:
    MAX, AVG, SUM, PNORM = "MAX", "AVG", "SUM", "PNORM"


class GradientNormalization:
    NONE = "None"
    RenormalizeL2PerLayer = "RenormalizeL2PerLayer"
    RenormalizeL2PerParamType = "RenormalizeL2PerParamType"
    ClipElementWiseAbsoluteValue = "ClipElementWiseAbsoluteValue"
    ClipL2PerLayer = "ClipL2PerLayer"
    ClipL2PerParamType = "ClipL2PerParamType"


class BackpropType:
    Standard, TruncatedBPTT = "Standard", "TruncatedBPTT"


class WorkspaceMode:
    NONE, ENABLED, SINGLE, SEPARATE = "NONE", "ENABLED", "SINGLE", "SEPARATE"


class CacheMode:
    NONE, HOST, DEVICE = "NONE", "HOST", "DEVICE"


class OptimizationAlgorithm:
    STOCHASTIC_GRADIENT_DESCENT = "STOCHASTIC_GRADIENT_DESCENT"
    LINE_GRADIENT_DESCENT = "LINE_GRADIENT_DESCENT"
    CONJUGATE_GRADIENT = "CONJUGATE_GRADIENT"
    LBFGS = "LBFGS"


class CNN2DFormat:
    NCHW, NHWC = "NCHW", "NHWC"


class RNNFormat:
    NCW, NWC = "NCW", "NWC"


class DataType:
    DOUBLE, FLOAT, HALF, BFLOAT16 = "DOUBLE", "FLOAT", "HALF", "BFLOAT16"
    INT, LONG, BOOL, UINT8, INT8, INT16 = "INT", "LONG", "BOOL", "UBYTE", "BYTE", "SHORT"

    _TORCH = {"DOUBLE": torch.float64, "FLOAT": torch.float32, "HALF": torch.float16, "BFLOAT16": torch.bfloat16,
              "INT": torch.int32, "LONG": torch.int64, "BOOL": torch.bool, "UBYTE": torch.uint8, "BYTE": torch.int8,
              "SHORT": torch.int16}

    @staticmethod
    def to_torch(dt):
        if isinstance(dt, torch.dtype):
            return dt
        return DataType._TORCH[str(dt).upper()]

    @staticmethod
    def from_torch(dt):
        for k, v in DataType._TORCH.items():
            if v == dt:
                return k
        raise ValueError(dt)


# ------------------------------------------------------------------ InputType
class InputType:
    FF, RNN, CNN, CNNFlat, CNN3D = "FF", "RNN", "CNN", "CNNFlat", "CNN3D"

    def __init__(self, kind, **kw):
        self.kind = kind
        self.size = kw.get("size")
        self.timeSeriesLength = kw.get("timeSeriesLength", -1)
        self.height, self.width, self.channels, self.depth = kw.get("height"), kw.get("width"), kw.get("channels"), kw.get("depth")
        self.format = kw.get("format", CNN2DFormat.NCHW)

    @staticmethod
    def feedForward(size):
        return InputType(InputType.FF, size=int(size))

    @staticmethod
    def recurrent(size, timeSeriesLength=-1, format=RNNFormat.NCW):
        return InputType(InputType.RNN, size=int(size), timeSeriesLength=int(timeSeriesLength), format=format)

    @staticmethod
    def convolutional(height, width, channels, format=CNN2DFormat.NCHW):
        return InputType(InputType.CNN, height=int(height), width=int(width), channels=int(channels), format=format)

    @staticmethod
    def convolutionalFlat(height, width, channels):
        return InputType(InputType.CNNFlat, height=int(height), width=int(width), channels=int(channels))

    @staticmethod
    def convolutional3D(depth, height, width, channels):
        return InputType(InputType.CNN3D, depth=int(depth), height=int(height), width=int(width), channels=int(channels))

    def arrayElementsPerExample(self):
        if self.kind == InputType.FF:
            return self.size
        if self.kind == InputType.RNN:
            return self.size * max(self.timeSeriesLength, 1)
        if self.kind == InputType.CNN3D:
            return self.depth * self.height * self.width * self.channels
        return self.height * self.width * self.channels

    def getFlattenedSize(self):
        return self.arrayElementsPerExample()

    def getShape(self, mb=1):
        if self.kind == InputType.FF:
            return [mb, self.size]
        if self.kind == InputType.RNN:
            return [mb, self.size, self.timeSeriesLength]
        if self.kind == InputType.CNN:
            return [mb, self.channels, self.height, self.width]
        if self.kind == InputType.CNNFlat:
            return [mb, self.channels * self.height * self.width]
        return [mb, self.channels, self.depth, self.height, self.width]

    def to_json(self):
        pk = "org.deeplearning4j.nn.conf.inputs.InputType$"
        if self.kind == InputType.FF:
            return {"@class": pk + "InputTypeFeedForward", "size": self.size}
        if self.kind == InputType.RNN:
            return {"@class": pk + "InputTypeRecurrent", "size": self.size, "timeSeriesLength": self.timeSeriesLength, "format": self.format}
        if self.kind == InputType.CNN:
            return {"@class": pk + "InputTypeConvolutional", "height": self.height, "width": self.width, "channels": self.channels, "format": self.format}
        if self.kind == InputType.CNNFlat:
            return {"@class": pk + "InputTypeConvolutionalFlat", "height": self.height, "width": self.width, "depth": self.channels}
        return {"@class": pk + "InputTypeConvolutional3D", "depth": self.depth, "height": self.height, "width": self.width, "channels": self.channels}

    @staticmethod
    def from_json(d):
        if d is None:
            return None
        c = d["@class"]
        if c.endswith("FeedForward"):
            return InputType.feedForward(d["size"])
        if c.endswith("Recurrent"):
            return InputType.recurrent(d["size"], d.get("timeSeriesLength", -1), d.get("format", RNNFormat.NCW))
        if c.endswith("ConvolutionalFlat"):
            return InputType.convolutionalFlat(d["height"], d["width"], d.get("depth", d.get("channels")))
        if c.endswith("Convolutional3D"):
            return InputType.convolutional3D(d["depth"], d["height"], d["width"], d["channels"])
        return InputType.convolutional(d["height"], d["width"], d.get("channels", d.get("depth")), d.get("format", CNN2DFormat.NCHW))

    def __eq__(self, o):
        return isinstance(o, InputType) and self.__dict__ == o.__dict__

    def __repr__(self):
        if self.kind == InputType.FF:
            return "InputTypeFeedForward(%d)" % self.size
        if self.kind == InputType.RNN:
            return "InputTypeRecurrent(%d,T=%d)" % (self.size, self.timeSeriesLength)
        if self.kind == InputType.CNN3D:
            return "InputTypeConvolutional3D(d=%d,h=%d,w=%d,c=%d)" % (self.depth, self.height, self.width, self.channels)
        return "InputType%s(h=%d,w=%d,c=%d)" % (self.kind, self.height, self.width, self.channels)


# ------------------------------------------------------------------ distributions
_DPKG = "org.deeplearning4j.nn.conf.distribution."


class Distribution:
    JAVA_CLASS = None

    def sample(self, shape, gen, dtype=torch.float32):
        raise NotImplementedError

    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        d.update(self.__dict__)
        return d

    def __eq__(self, o):
        return type(self) is type(o) and self.__dict__ == o.__dict__

    def __hash__(self):
        return hash(type(self).__name__)


class NormalDistribution(Distribution):
    JAVA_CLASS = _DPKG + "NormalDistribution"

    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = float(mean), float(std)

    def sample(self, shape, gen, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype).normal_(self.mean, self.std, generator=gen)


GaussianDistribution = NormalDistribution


class UniformDistribution(Distribution):
    JAVA_CLASS = _DPKG + "UniformDistribution"

    def __init__(self, lower=0.0, upper=1.0):
        self.lower, self.upper = float(lower), float(upper)

    def sample(self, shape, gen, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype).uniform_(self.lower, self.upper, generator=gen)


class TruncatedNormalDistribution(Distribution):
    JAVA_CLASS = _DPKG + "TruncatedNormalDistribution"

    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = float(mean), float(std)

    def sample(self, shape, gen, dtype=torch.float32):
        # resample values outside 2 sigma (libnd4j TruncatedNormalDistribution semantics)
        t = torch.empty(shape, dtype=dtype).normal_(0.0, 1.0, generator=gen)
        for _ in range(8):
            bad = t.abs() > 2.0
            if not bool(bad.any()):
                break
            t[bad] = torch.empty(int(bad.sum()), dtype=dtype).normal_(0.0, 1.0, generator=gen)
        return t.clamp_(-2.0, 2.0) * self.std + self.mean


class LogNormalDistribution(Distribution):
    JAVA_CLASS = _DPKG + "LogNormalDistribution"

    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = float(mean), float(std)

    def sample(self, shape, gen, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype).log_normal_(self.mean, self.std, generator=gen)


class BinomialDistribution(Distribution):
    JAVA_CLASS = _DPKG + "BinomialDistribution"

    def __init__(self, numberOfTrials=1, probabilityOfSuccess=0.5):
        self.numberOfTrials, self.probabilityOfSuccess = int(numberOfTrials), float(probabilityOfSuccess)

    def sample(self, shape, gen, dtype=torch.float32):
        p = torch.full(tuple(shape) + (self.numberOfTrials,), self.probabilityOfSuccess)
        return torch.bernoulli(p, generator=gen).sum(-1).to(dtype)


class ConstantDistribution(Distribution):
    JAVA_CLASS = _DPKG + "ConstantDistribution"

    def __init__(self, value=0.0):
        self.value = float(value)

    def sample(self, shape, gen, dtype=torch.float32):
        return torch.full(shape, self.value, dtype=dtype)


class OrthogonalDistribution(Distribution):
    JAVA_CLASS = _DPKG + "OrthogonalDistribution"

    def __init__(self, gain=1.0):
        self.gain = float(gain)

    def sample(self, shape, gen, dtype=torch.float32):
        rows = shape[0]
        cols = int(math.prod(shape[1:])) if len(shape) > 1 else 1
        a = torch.empty(max(rows, cols), min(rows, cols), dtype=torch.float64).normal_(0, 1, generator=gen)
        q, r = torch.linalg.qr(a)
        q = q * torch.sign(torch.diagonal(r))
        if rows < cols:
            q = q.t()
        return (self.gain * q[:rows, :cols]).reshape(shape).to(dtype)


_DISTS = {c.JAVA_CLASS: c for c in (NormalDistribution, UniformDistribution, TruncatedNormalDistribution,
                                    LogNormalDistribution, BinomialDistribution, ConstantDistribution, OrthogonalDistribution)}
_DISTS[_DPKG + "GaussianDistribution"] = NormalDistribution


def distribution_from_json(d):
    if d is None:
        return None
    cls = _DISTS[d["@class"]]
    return cls(**{k: v for k, v in d.items() if k != "@class"})


# ------------------------------------------------------------------ weight init
_WPKG = "org.deeplearning4j.nn.weights."


class WeightInit:
    DISTRIBUTION = "DISTRIBUTION"; ZERO = "ZERO"; ONES = "ONES"; SIGMOID_UNIFORM = "SIGMOID_UNIFORM"; NORMAL = "NORMAL"
    LECUN_NORMAL = "LECUN_NORMAL"; UNIFORM = "UNIFORM"; XAVIER = "XAVIER"; XAVIER_UNIFORM = "XAVIER_UNIFORM"
    XAVIER_FAN_IN = "XAVIER_FAN_IN"; XAVIER_LEGACY = "XAVIER_LEGACY"; RELU = "RELU"; RELU_UNIFORM = "RELU_UNIFORM"
    IDENTITY = "IDENTITY"; LECUN_UNIFORM = "LECUN_UNIFORM"; VAR_SCALING_NORMAL_FAN_IN = "VAR_SCALING_NORMAL_FAN_IN"
    VAR_SCALING_NORMAL_FAN_OUT = "VAR_SCALING_NORMAL_FAN_OUT"; VAR_SCALING_NORMAL_FAN_AVG = "VAR_SCALING_NORMAL_FAN_AVG"
    VAR_SCALING_UNIFORM_FAN_IN = "VAR_SCALING_UNIFORM_FAN_IN"; VAR_SCALING_UNIFORM_FAN_OUT = "VAR_SCALING_UNIFORM_FAN_OUT"
    VAR_SCALING_UNIFORM_FAN_AVG = "VAR_SCALING_UNIFORM_FAN_AVG"


_W_CLASS = {
    "ZERO": "WeightInitConstant", "ONES": "WeightInitConstant", "SIGMOID_UNIFORM": "WeightInitSigmoidUniform",
    "NORMAL": "WeightInitNormal", "LECUN_NORMAL": "WeightInitLecunNormal", "UNIFORM": "WeightInitUniform",
    "XAVIER": "WeightInitXavier", "XAVIER_UNIFORM": "WeightInitXavierUniform", "XAVIER_FAN_IN": "WeightInitXavierFanIn",
    "XAVIER_LEGACY": "WeightInitXavierLegacy", "RELU": "WeightInitRelu", "RELU_UNIFORM": "WeightInitReluUniform",
    "IDENTITY": "WeightInitIdentity", "LECUN_UNIFORM": "WeightInitLecunUniform",
    "VAR_SCALING_NORMAL_FAN_IN": "WeightInitVarScalingNormalFanIn", "VAR_SCALING_NORMAL_FAN_OUT": "WeightInitVarScalingNormalFanOut",
    "VAR_SCALING_NORMAL_FAN_AVG": "WeightInitVarScalingNormalFanAvg", "VAR_SCALING_UNIFORM_FAN_IN": "WeightInitVarScalingUniformFanIn",
    "VAR_SCALING_UNIFORM_FAN_OUT": "WeightInitVarScalingUniformFanOut", "VAR_SCALING_UNIFORM_FAN_AVG": "WeightInitVarScalingUniformFanAvg",
    "DISTRIBUTION": "WeightInitDistribution",
}
_W_BY_CLASS = {}
for _k, _v in _W_CLASS.items():
    _W_BY_CLASS.setdefault(_WPKG + _v, _k)


class IWeightInit:
    """Weight initialiser; ``scheme`` is a WeightInit enum name, ``distribution`` for DISTRIBUTION."""

    def __init__(self, scheme="XAVIER", distribution: Optional[Distribution] = None, value: float = 0.0):
        self.scheme, self.distribution, self.value = str(scheme).upper(), distribution, value
        if self.scheme == "ONES":
            self.value = 1.0

    def init(self, fan_in, fan_out, shape, gen, dtype=torch.float32):
        s = self.scheme
        shape = tuple(int(x) for x in shape)
        n = lambda std: torch.empty(shape, dtype=dtype).normal_(0.0, std, generator=gen)
        u = lambda a: torch.empty(shape, dtype=dtype).uniform_(-a, a, generator=gen)
        tn = lambda std: TruncatedNormalDistribution(0.0, std).sample(shape, gen, dtype)
        if s == "DISTRIBUTION":
            return self.distribution.sample(shape, gen, dtype)
        if s in ("ZERO", "ONES"):
            return torch.full(shape, float(self.value), dtype=dtype)
        if s == "XAVIER":
            return n(math.sqrt(2.0 / (fan_in + fan_out)))
        if s == "XAVIER_UNIFORM":
            return u(math.sqrt(6.0 / (fan_in + fan_out)))
        if s == "XAVIER_FAN_IN":
            return n(math.sqrt(1.0 / fan_in))
        if s == "XAVIER_LEGACY":
            return n(math.sqrt(1.0 / (shape[0] + shape[1])))
        if s == "RELU":
            return n(math.sqrt(2.0 / fan_in))
        if s == "RELU_UNIFORM":
            return u(math.sqrt(6.0 / fan_in))
        if s == "SIGMOID_UNIFORM":
            return u(4.0 * math.sqrt(6.0 / (fan_in + fan_out)))
        if s == "UNIFORM":
            return u(1.0 / math.sqrt(fan_in))
        if s == "LECUN_UNIFORM":
            return u(3.0 / math.sqrt(fan_in))
        if s in ("NORMAL", "LECUN_NORMAL"):
            return n(math.sqrt(1.0 / fan_in))
        if s == "IDENTITY":
            if len(shape) == 2 and shape[0] == shape[1]:
                return torch.eye(shape[0], dtype=dtype)
            t = torch.zeros(shape, dtype=dtype)
            if len(shape) >= 3 and shape[0] == shape[1]:
                ctr = tuple(x // 2 for x in shape[2:])
                for i in range(shape[0]):
                    t[(i, i) + ctr] = 1.0
                return t
            raise ValueError("IDENTITY weight init needs square shape, got %s" % (shape,))
        if s.startswith("VAR_SCALING"):
            fan = {"IN": fan_in, "OUT": fan_out, "AVG": (fan_in + fan_out) / 2.0}[s.rsplit("_", 1)[1]]
            if "NORMAL" in s:
                return tn(math.sqrt(1.0 / fan))
            return u(math.sqrt(3.0 / fan))
        raise ValueError("unknown weight init " + s)

    def to_json(self):
        d = {"@class": _WPKG + _W_CLASS[self.scheme]}
        if self.scheme == "DISTRIBUTION":
            d["distribution"] = self.distribution.to_json()
        if self.scheme in ("ZERO", "ONES"):
            d["value"] = self.value
        return d

    @staticmethod
    def from_json(d):
        if d is None:
            return None
        c = d["@class"]
        scheme = _W_BY_CLASS[c]
        if scheme == "DISTRIBUTION":
            return IWeightInit("DISTRIBUTION", distribution_from_json(d["distribution"]))
        if c.endswith("WeightInitConstant"):
            v = d.get("value", 0.0)
            return IWeightInit("ONES" if v == 1.0 else "ZERO", value=v)
        return IWeightInit(scheme)

    def __eq__(self, o):
        return isinstance(o, IWeightInit) and (self.scheme, self.distribution, self.value) == (o.scheme, o.distribution, o.value)

    def __hash__(self):
        return hash(self.scheme)

    def __repr__(self):
        return "WeightInit.%s" % self.scheme


def WeightInitDistribution(dist):
    return IWeightInit("DISTRIBUTION", dist)


def get_weight_init(w, dist=None):
    if w is None:
        return None
    if isinstance(w, IWeightInit):
        return w
    if isinstance(w, Distribution):
        return IWeightInit("DISTRIBUTION", w)
    if isinstance(w, dict):
        return IWeightInit.from_json(w)
    return IWeightInit(w, dist)
