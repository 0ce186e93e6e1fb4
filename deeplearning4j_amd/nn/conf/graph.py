"""GraphVertex configurations (reference: deeplearning4j-nn/.../nn/conf/graph/*.java, impl
nn/graph/vertex/impl/*.java). Runtime implementations live in nn/graph/vertices.py."""
from __future__ import annotations

from .base import InputType
from .layers import Layer
from .preprocessors import preprocessor_from_json

_G = "org.deeplearning4j.nn.conf.graph."


class GraphVertex:
    JAVA_CLASS = None
    FIELDS = ()

    def numParams(self, backprop=False):
        return 0

    def getOutputType(self, layerIndex, *inputTypes):
        return inputTypes[0]

    def minVertexInputs(self):
        return 1

    def maxVertexInputs(self):
        return 1

    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        for f in self.FIELDS:
            v = getattr(self, f)
            d[f] = v.to_json() if hasattr(v, "to_json") else v
        return d

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def __eq__(self, o):
        return type(self) is type(o) and self.to_json() == o.to_json()

    def __hash__(self):
        return hash(type(self).__name__)


class LayerVertex(GraphVertex):
    JAVA_CLASS = _G + "LayerVertex"

    def __init__(self, layerConf, preProcessor=None, outputVertex=False):
        self.layerConf, self.preProcessor, self.outputVertex = layerConf, preProcessor, outputVertex

    @property
    def layer(self) -> Layer:
        return self.layerConf.layer

    def numParams(self, backprop=False):
        return self.layer.numParams()

    def getOutputType(self, layerIndex, *inputTypes):
        it = inputTypes[0]
        if self.preProcessor is not None:
            it = self.preProcessor.getOutputType(it)
        return self.layer.getOutputType(layerIndex, it)

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "layerConf": self.layerConf.to_json(),
                "preProcessor": self.preProcessor.to_json() if self.preProcessor else None,
                "outputVertex": self.outputVertex}


class MergeVertex(GraphVertex):
    JAVA_CLASS = _G + "MergeVertex"
    FIELDS = ("mergeAxis",)

    def __init__(self, mergeAxis=1):
        self.mergeAxis = mergeAxis

    def maxVertexInputs(self):
        return 1 << 30

    def getOutputType(self, layerIndex, *its):
        f = its[0]
        if len(its) == 1:
            return f
        if f.kind == InputType.FF:
            return InputType.feedForward(sum(i.size for i in its))
        if f.kind == InputType.RNN:
            return InputType.recurrent(sum(i.size for i in its), f.timeSeriesLength)
        if f.kind == InputType.CNN3D:
            return InputType.convolutional3D(f.depth, f.height, f.width, sum(i.channels for i in its))
        return InputType.convolutional(f.height, f.width, sum(i.channels for i in its), f.format)


class ElementWiseVertex(GraphVertex):
    JAVA_CLASS = _G + "ElementWiseVertex"
    FIELDS = ("op",)

    class Op:
        Add, Subtract, Product, Average, Max = "Add", "Subtract", "Product", "Average", "Max"

    def __init__(self, op="Add"):
        self.op = op

    def minVertexInputs(self):
        return 2

    def maxVertexInputs(self):
        return 2 if self.op == "Subtract" else 1 << 30


class SubsetVertex(GraphVertex):
    JAVA_CLASS = _G + "SubsetVertex"
    FIELDS = ("from_", "to")

    def __init__(self, from_=0, to=0, **kw):
        self.from_ = kw.get("from", from_)
        self.to = to

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "from": self.from_, "to": self.to}

    def getOutputType(self, layerIndex, *its):
        n = self.to - self.from_ + 1
        f = its[0]
        if f.kind == InputType.FF:
            return InputType.feedForward(n)
        if f.kind == InputType.RNN:
            return InputType.recurrent(n, f.timeSeriesLength)
        return InputType.convolutional(f.height, f.width, n, f.format)


class StackVertex(GraphVertex):
    JAVA_CLASS = _G + "StackVertex"

    def maxVertexInputs(self):
        return 1 << 30


class UnstackVertex(GraphVertex):
    JAVA_CLASS = _G + "UnstackVertex"
    FIELDS = ("from_", "stackSize")

    def __init__(self, from_=0, stackSize=1, **kw):
        self.from_ = kw.get("from", from_)
        self.stackSize = stackSize

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "from": self.from_, "stackSize": self.stackSize}


class ScaleVertex(GraphVertex):
    JAVA_CLASS = _G + "ScaleVertex"
    FIELDS = ("scaleFactor",)

    def __init__(self, scaleFactor=1.0):
        self.scaleFactor = float(scaleFactor)


class ShiftVertex(GraphVertex):
    JAVA_CLASS = _G + "ShiftVertex"
    FIELDS = ("shiftFactor",)

    def __init__(self, shiftFactor=0.0):
        self.shiftFactor = float(shiftFactor)


class L2Vertex(GraphVertex):
    JAVA_CLASS = _G + "L2Vertex"
    FIELDS = ("eps",)

    def __init__(self, eps=1e-8):
        self.eps = eps

    def minVertexInputs(self):
        return 2

    def maxVertexInputs(self):
        return 2

    def getOutputType(self, layerIndex, *its):
        return InputType.feedForward(1)


class L2NormalizeVertex(GraphVertex):
    JAVA_CLASS = _G + "L2NormalizeVertex"
    FIELDS = ("dimension", "eps")

    def __init__(self, dimension=None, eps=1e-8):
        self.dimension, self.eps = dimension, eps


class ReshapeVertex(GraphVertex):
    JAVA_CLASS = _G + "ReshapeVertex"
    FIELDS = ("reshapeOrder", "newShape", "maskShape")

    def __init__(self, *newShape, reshapeOrder="c", maskShape=None, **kw):
        if "newShape" in kw:
            newShape = kw["newShape"]
        if len(newShape) == 1 and isinstance(newShape[0], (list, tuple)):
            newShape = newShape[0]
        self.newShape, self.reshapeOrder, self.maskShape = [int(x) for x in newShape], reshapeOrder, maskShape

    def getOutputType(self, layerIndex, *its):
        s = self.newShape
        if len(s) == 2:
            return InputType.feedForward(s[1])
        if len(s) == 3:
            return InputType.recurrent(s[1], s[2])
        if len(s) == 4:
            return InputType.convolutional(s[2], s[3], s[1])
        return its[0]


class PreprocessorVertex(GraphVertex):
    JAVA_CLASS = _G + "PreprocessorVertex"

    def __init__(self, preProcessor=None):
        self.preProcessor = preprocessor_from_json(preProcessor) if isinstance(preProcessor, dict) else preProcessor

    def getOutputType(self, layerIndex, *its):
        return self.preProcessor.getOutputType(its[0])

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "preProcessor": self.preProcessor.to_json()}


class PoolHelperVertex(GraphVertex):
    JAVA_CLASS = _G + "PoolHelperVertex"

    def getOutputType(self, layerIndex, *its):
        f = its[0]
        return InputType.convolutional(f.height - 1, f.width - 1, f.channels, f.format)


class FrozenVertex(GraphVertex):
    JAVA_CLASS = _G + "FrozenVertex"

    def __init__(self, underlying=None):
        self.underlying = vertex_from_json(underlying) if isinstance(underlying, dict) else underlying

    def getOutputType(self, layerIndex, *its):
        return self.underlying.getOutputType(layerIndex, *its)

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "underlying": self.underlying.to_json()}


class LastTimeStepVertex(GraphVertex):
    JAVA_CLASS = _G + "rnn.LastTimeStepVertex"
    FIELDS = ("maskArrayInputName",)

    def __init__(self, maskArrayInputName=None):
        self.maskArrayInputName = maskArrayInputName

    def getOutputType(self, layerIndex, *its):
        return InputType.feedForward(its[0].size)


class DuplicateToTimeSeriesVertex(GraphVertex):
    JAVA_CLASS = _G + "rnn.DuplicateToTimeSeriesVertex"
    FIELDS = ("inputName",)

    def __init__(self, inputName=None):
        self.inputName = inputName

    def getOutputType(self, layerIndex, *its):
        return InputType.recurrent(its[0].size)


class ReverseTimeSeriesVertex(GraphVertex):
    JAVA_CLASS = _G + "rnn.ReverseTimeSeriesVertex"
    FIELDS = ("maskArrayInputName",)

    def __init__(self, maskArrayInputName=None):
        self.maskArrayInputName = maskArrayInputName


class _ParamVertexMixin:
    """Vertices that own parameters take the network-wide updater / weight init (GraphBuilder.build -> apply_global)."""
    iupdater = None
    weightInitFn = None

    def apply_global(self, g):
        import copy
        if self.iupdater is None and g.get("iupdater") is not None:
            self.iupdater = copy.deepcopy(g["iupdater"])
        if self.weightInitFn is None and g.get("weightInitFn") is not None:
            self.weightInitFn = copy.deepcopy(g["weightInitFn"])

    def getUpdaterByParam(self, spec):
        from ...nd4j.learning import NoOp
        return self.iupdater or NoOp()

    def getRegularizationByParam(self, spec):
        return []


class AttentionVertex(_ParamVertexMixin, GraphVertex):
    JAVA_CLASS = _G + "AttentionVertex"
    FIELDS = ("nInKeys", "nInValues", "nInQueries", "nOut", "headSize", "nHeads", "projectInput")

    def __init__(self, nInKeys=0, nInValues=0, nInQueries=0, nOut=0, headSize=0, nHeads=1, projectInput=True, weightInit="XAVIER"):
        self.nInKeys, self.nInValues, self.nInQueries, self.nOut = nInKeys, nInValues, nInQueries, nOut
        self.headSize, self.nHeads, self.projectInput, self.weightInit = headSize or (nOut // max(nHeads, 1)), nHeads, projectInput, weightInit

    def minVertexInputs(self):
        return 3

    def maxVertexInputs(self):
        return 3

    def numParams(self, backprop=False):
        if not self.projectInput:
            return 0
        h, hs = self.nHeads, self.headSize
        return h * hs * (self.nInQueries + self.nInKeys + self.nInValues) + h * hs * self.nOut

    def getOutputType(self, layerIndex, *its):
        return InputType.recurrent(self.nOut if self.projectInput else self.nInValues, its[0].timeSeriesLength)


class DotProductAttentionVertex(GraphVertex):
    JAVA_CLASS = _G + "DotProductAttentionVertex"
    FIELDS = ("scaleFactor", "dropoutProbability", "useCausalMask")

    def __init__(self, scaleFactor=0.0, dropoutProbability=0.0, useCausalMask=False):
        self.scaleFactor, self.dropoutProbability, self.useCausalMask = scaleFactor, dropoutProbability, useCausalMask

    def minVertexInputs(self):
        return 2

    def maxVertexInputs(self):
        return 3


class SameDiffVertex(_ParamVertexMixin, GraphVertex):
    """User-defined vertex with parameters (reference: nn/conf/layers/samediff/SameDiffVertex.java): subclass and implement
    ``defineParameters() -> {name: shape}``, ``defineVertex(inputs: list, params: dict) -> tensor`` (torch ops; autograd supplies
    the backward) and ``getOutputType``."""
    JAVA_CLASS = "org.deeplearning4j.nn.conf.layers.samediff.SameDiffVertex"

    def defineParameters(self):
        return {}

    def defineVertex(self, inputs, params):
        raise NotImplementedError

    def minVertexInputs(self):
        return 1

    def maxVertexInputs(self):
        return 1 << 30

    def param_specs(self):
        from .layers import ParamSpec
        return [ParamSpec(k, tuple(shp), "c", shp[0] if shp else 1, shp[-1] if shp else 1, "b" if k.startswith("b") else "W")
                for k, shp in self.defineParameters().items()]

    def numParams(self, backprop=False):
        return sum(sp.numel for sp in self.param_specs())

    def getOutputType(self, layerIndex, *its):
        return its[0]

    def to_json(self):
        return {"@class": self.JAVA_CLASS, "pythonClass": type(self).__module__ + "." + type(self).__qualname__}


class SameDiffLambdaVertex(SameDiffVertex):
    """Parameter-free vertex from a callable over its inputs (SameDiffLambdaVertex.java)."""
    JAVA_CLASS = "org.deeplearning4j.nn.conf.layers.samediff.SameDiffLambdaVertex"

    def __init__(self, fn=None, outputTypeFn=None):
        self.fn, self.outputTypeFn = fn, outputTypeFn

    def defineVertex(self, inputs, params):
        return self.fn(*inputs)

    def getOutputType(self, layerIndex, *its):
        return self.outputTypeFn(*its) if self.outputTypeFn else its[0]


_VERTS = {c.JAVA_CLASS: c for c in (MergeVertex, ElementWiseVertex, SubsetVertex, StackVertex, UnstackVertex, ScaleVertex,
                                    ShiftVertex, L2Vertex, L2NormalizeVertex, ReshapeVertex, PreprocessorVertex,
                                    PoolHelperVertex, FrozenVertex, LastTimeStepVertex, DuplicateToTimeSeriesVertex,
                                    ReverseTimeSeriesVertex, AttentionVertex, DotProductAttentionVertex)}


def vertex_from_json(d):
    c = d["@class"]
    if c == LayerVertex.JAVA_CLASS:
        from .configuration import NeuralNetConfiguration
        return LayerVertex(NeuralNetConfiguration.from_json_obj(d["layerConf"]),
                           preprocessor_from_json(d.get("preProcessor")), d.get("outputVertex", False))
    cls = _VERTS[c]
    kw = {k: v for k, v in d.items() if k != "@class"}
    return cls(**kw)
