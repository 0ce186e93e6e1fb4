"""NeuralNetConfiguration / MultiLayerConfiguration / ComputationGraphConfiguration.

Reference: deeplearning4j-nn/.../nn/conf/NeuralNetConfiguration.java (Builder, ListBuilder),
MultiLayerConfiguration.java, ComputationGraphConfiguration.java (GraphBuilder, topological
validation, addPreProcessors). JSON/YAML layout follows the Jackson output so
``configuration.json`` inside ModelSerializer zips round-trips.

MI355X extension (not in the reference API, which rejects bf16 networks —
NeuralNetConfiguration.java:1149-1154): ``computeDataType(DataType.BFLOAT16)`` keeps fp32 master
parameters/updater state and runs activations + weight replicas in bf16 on the MFMA path.
"""
from __future__ import annotations

import copy
import json
from collections import OrderedDict
from typing import Dict, List, Optional

from ...nd4j.activations import get_activation
from ...nd4j.learning import IUpdater, Sgd, updater_from_json
from .base import (InputType, BackpropType, WorkspaceMode, CacheMode, OptimizationAlgorithm, ConvolutionMode,
                   GradientNormalization, get_weight_init, DataType)
from .layers import Layer, BaseLayer, Dropout, IDropout, dropout_from_json, INHERITABLE, BaseOutputLayer, LossLayer
from .preprocessors import preprocessor_from_json
from . import graph as G


class NeuralNetConfiguration:
    """Per-layer configuration holder (one per layer, as in DL4J) + global defaults."""

    def __init__(self, layer: Optional[Layer] = None, **kw):
        self.layer = layer
        self.seed = kw.get("seed", 12345)
        self.miniBatch = kw.get("miniBatch", True)
        self.minimize = kw.get("minimize", True)
        self.maxNumLineSearchIterations = kw.get("maxNumLineSearchIterations", 5)
        self.optimizationAlgo = kw.get("optimizationAlgo", OptimizationAlgorithm.STOCHASTIC_GRADIENT_DESCENT)
        self.cacheMode = kw.get("cacheMode", CacheMode.NONE)
        self.dataType = kw.get("dataType", DataType.FLOAT)
        self.iterationCount = kw.get("iterationCount", 0)
        self.epochCount = kw.get("epochCount", 0)
        self.variables = kw.get("variables", [])
        self.stepFunction = None

    def clone(self):
        return copy.deepcopy(self)

    def getLayer(self):
        return self.layer

    def to_json(self):
        return {"cacheMode": self.cacheMode, "dataType": self.dataType, "epochCount": self.epochCount,
                "iterationCount": self.iterationCount, "layer": self.layer.to_json() if self.layer else None,
                "maxNumLineSearchIterations": self.maxNumLineSearchIterations, "miniBatch": self.miniBatch,
                "minimize": self.minimize, "optimizationAlgo": self.optimizationAlgo, "seed": self.seed,
                "stepFunction": None, "variables": [s.key for s in self.layer.param_specs()] if self.layer else []}

    def toJson(self):
        return json.dumps(self.to_json(), indent=2, sort_keys=True)

    @staticmethod
    def from_json_obj(d, _upgraded=False):
        if not _upgraded:
            from . import legacy_json
            d = legacy_json.upgrade(d)            # wrapper-object subtypes / enum-era fields of old releases (idempotent)
        layer = Layer.from_json(d["layer"]) if d.get("layer") else None
        kw = {k: d[k] for k in ("seed", "miniBatch", "minimize", "maxNumLineSearchIterations", "optimizationAlgo",
                                "cacheMode", "dataType", "iterationCount", "epochCount", "variables") if k in d}
        return NeuralNetConfiguration(layer, **kw)

    @staticmethod
    def fromJson(s):
        return NeuralNetConfiguration.from_json_obj(json.loads(s))

    # ------------------------------------------------------------------
    class Builder:
        def __init__(self):
            self._g = {f: None for f in INHERITABLE}
            self._g.update(activationFn=get_activation("SIGMOID"), weightInitFn=get_weight_init("XAVIER"), biasInit=0.0,
                           gainInit=1.0, iupdater=Sgd(), convolutionMode=ConvolutionMode.Truncate,
                           gradientNormalization=GradientNormalization.NONE, gradientNormalizationThreshold=1.0)
            self._seed = 12345
            self._miniBatch = True
            self._minimize = True
            self._optimizationAlgo = OptimizationAlgorithm.STOCHASTIC_GRADIENT_DESCENT
            self._maxLS = 5
            self._cacheMode = CacheMode.NONE
            self._trainWs = WorkspaceMode.ENABLED
            self._infWs = WorkspaceMode.ENABLED
            self._dataType = DataType.FLOAT
            self._computeDataType = None
            self._cudnnAlgoMode = "PREFER_FASTEST"
            self._layer = None

        # --- global hyper-parameters (fluent)
        def seed(self, s):
            self._seed = int(s); return self

        def activation(self, a):
            self._g["activationFn"] = get_activation(a); return self

        def weightInit(self, w, dist=None):
            self._g["weightInitFn"] = get_weight_init(w, dist); return self

        def dist(self, d):
            self._g["weightInitFn"] = get_weight_init(d); return self

        def biasInit(self, b):
            self._g["biasInit"] = float(b); return self

        def gainInit(self, b):
            self._g["gainInit"] = float(b); return self

        def updater(self, u):
            self._g["iupdater"] = u; return self

        def biasUpdater(self, u):
            self._g["biasUpdater"] = u; return self

        def l1(self, v):
            self._g["l1"] = float(v); return self

        def l2(self, v):
            self._g["l2"] = float(v); return self

        def l1Bias(self, v):
            self._g["l1Bias"] = float(v); return self

        def l2Bias(self, v):
            self._g["l2Bias"] = float(v); return self

        def weightDecay(self, v, applyLR=True):
            from ...nd4j.learning import WeightDecay
            self._g["weightDecay"] = WeightDecay(v, applyLR); return self

        def weightDecayBias(self, v, applyLR=True):
            from ...nd4j.learning import WeightDecay
            self._g["weightDecayBias"] = WeightDecay(v, applyLR); return self

        def dropOut(self, d):
            self._g["idropout"] = d if isinstance(d, IDropout) else (Dropout(float(d)) if d else None); return self

        def weightNoise(self, w):
            self._g["weightNoise"] = w; return self

        def constrainWeights(self, *c):
            self._g["constraints"] = list(c); return self

        constrainAllParameters = constrainWeights
        constrainBias = constrainWeights

        def gradientNormalization(self, g):
            self._g["gradientNormalization"] = g; return self

        def gradientNormalizationThreshold(self, t):
            self._g["gradientNormalizationThreshold"] = float(t); return self

        def convolutionMode(self, m):
            self._g["convolutionMode"] = m; return self

        def cudnnAlgoMode(self, m):
            self._cudnnAlgoMode = m; return self

        def miniBatch(self, b):
            self._miniBatch = bool(b); return self

        def minimize(self, b):
            self._minimize = bool(b); return self

        def optimizationAlgo(self, a):
            self._optimizationAlgo = a; return self

        def maxNumLineSearchIterations(self, n):
            self._maxLS = int(n); return self

        def cacheMode(self, m):
            self._cacheMode = m; return self

        def trainingWorkspaceMode(self, m):
            self._trainWs = m; return self

        def inferenceWorkspaceMode(self, m):
            self._infWs = m; return self

        def dataType(self, dt):
            dt = str(dt).upper()
            if dt not in (DataType.DOUBLE, DataType.FLOAT, DataType.HALF):
                raise ValueError("Network dataType must be DOUBLE, FLOAT or HALF (got %s); use computeDataType("
                                 "BFLOAT16) for bf16 MFMA execution" % dt)
            self._dataType = dt; return self

        def computeDataType(self, dt):
            self._computeDataType = None if dt is None else str(dt).upper(); return self

        def layer(self, l):
            self._layer = l; return self

        # --- terminal builders
        def _nnc(self, layer):
            if layer is not None:
                layer = layer.clone() if False else layer
                layer.apply_global(self._g)
            return NeuralNetConfiguration(layer, seed=self._seed, miniBatch=self._miniBatch, minimize=self._minimize,
                                          maxNumLineSearchIterations=self._maxLS, optimizationAlgo=self._optimizationAlgo,
                                          cacheMode=self._cacheMode, dataType=self._dataType)

        def build(self):
            return self._nnc(self._layer)

        def list(self, *layers):
            lb = ListBuilder(self)
            for i, l in enumerate(layers):
                lb.layer(i, l)
            return lb

        def graphBuilder(self):
            return GraphBuilder(self)


class ListBuilder:
    def __init__(self, gb: "NeuralNetConfiguration.Builder"):
        self._gb = gb
        self._layers: Dict[int, Layer] = {}
        self._pre: Dict[int, object] = {}
        self._inputType = None
        self._backpropType = BackpropType.Standard
        self._tbpttFwd = 20
        self._tbpttBack = 20
        self._validateOutput = True
        self._overrideNinUponBuild = True

    def layer(self, *args):
        if len(args) == 1:
            self._layers[len(self._layers)] = args[0]
        else:
            self._layers[int(args[0])] = args[1]
        return self

    def inputPreProcessor(self, idx, p):
        self._pre[int(idx)] = p; return self

    def setInputType(self, it):
        self._inputType = it; return self

    inputType = setInputType

    def backpropType(self, t):
        self._backpropType = t; return self

    def tBPTTLength(self, n):
        self._tbpttFwd = self._tbpttBack = int(n); return self

    def tBPTTForwardLength(self, n):
        self._tbpttFwd = int(n); return self

    def tBPTTBackwardLength(self, n):
        self._tbpttBack = int(n); return self

    def validateOutputLayerConfig(self, b):
        self._validateOutput = bool(b); return self

    def overrideNinUponBuild(self, b):
        self._overrideNinUponBuild = bool(b); return self

    def build(self) -> "MultiLayerConfiguration":
        n = len(self._layers)
        if sorted(self._layers.keys()) != list(range(n)):
            raise ValueError("Layer indices must be 0..n-1 without gaps")
        confs = [self._gb._nnc(self._layers[i]) for i in range(n)]
        conf = MultiLayerConfiguration(confs, dict(self._pre), backpropType=self._backpropType, tbpttFwdLength=self._tbpttFwd,
                                       tbpttBackLength=self._tbpttBack, trainingWorkspaceMode=self._gb._trainWs,
                                       inferenceWorkspaceMode=self._gb._infWs, cacheMode=self._gb._cacheMode,
                                       dataType=self._gb._dataType, computeDataType=self._gb._computeDataType,
                                       validateOutputLayerConfig=self._validateOutput, inputType=self._inputType)
        if self._inputType is not None:
            conf._apply_input_type(self._inputType, self._overrideNinUponBuild)
        elif n and 0 not in self._pre:
            # no InputType given: infer it from the first layer's nIn where that is unambiguous
            # (MultiLayerConfiguration.Builder.build, reference MultiLayerConfiguration.java:700-735)
            from .layers import BaseRecurrentLayer, FeedForwardLayer, ConvolutionLayer, RnnOutputLayer, EmbeddingSequenceLayer
            l0 = self._layers[0]
            nin = getattr(l0, "nIn", 0) or 0
            if nin > 0 and isinstance(l0, FeedForwardLayer) and not isinstance(l0, ConvolutionLayer):
                rnn = isinstance(l0, (BaseRecurrentLayer, RnnOutputLayer)) and not isinstance(l0, EmbeddingSequenceLayer)
                it = InputType.recurrent(nin) if rnn else InputType.feedForward(nin)
                conf._apply_input_type(it, False)
        return conf


class MultiLayerConfiguration:
    def __init__(self, confs: List[NeuralNetConfiguration], inputPreProcessors=None, **kw):
        self.confs = confs
        self.inputPreProcessors = inputPreProcessors or {}
        self.backpropType = kw.get("backpropType", BackpropType.Standard)
        self.tbpttFwdLength = kw.get("tbpttFwdLength", 20)
        self.tbpttBackLength = kw.get("tbpttBackLength", 20)
        self.trainingWorkspaceMode = kw.get("trainingWorkspaceMode", WorkspaceMode.ENABLED)
        self.inferenceWorkspaceMode = kw.get("inferenceWorkspaceMode", WorkspaceMode.ENABLED)
        self.cacheMode = kw.get("cacheMode", CacheMode.NONE)
        self.dataType = kw.get("dataType", DataType.FLOAT)
        self.computeDataType = kw.get("computeDataType")
        self.validateOutputLayerConfig = kw.get("validateOutputLayerConfig", True)
        self.iterationCount = kw.get("iterationCount", 0)
        self.epochCount = kw.get("epochCount", 0)
        self.inputType = kw.get("inputType")

    def getConf(self, i):
        return self.confs[i]

    def getInputPreProcess(self, i):
        return self.inputPreProcessors.get(i)

    def clone(self):
        return copy.deepcopy(self)

    def _apply_input_type(self, it, override=True):
        cur = it
        for i, c in enumerate(self.confs):
            l = c.layer
            if i not in self.inputPreProcessors:
                p = l.getPreProcessorForInputType(cur)
                if p is not None:
                    self.inputPreProcessors[i] = p
            if i in self.inputPreProcessors:
                cur = self.inputPreProcessors[i].getOutputType(cur)
            l.setNIn(cur, override)
            cur = l.getOutputType(i, cur)

    def getMemoryReport(self, inputType=None):
        from .memory import memory_report_mln
        return memory_report_mln(self, inputType)

    def getLayerActivationTypes(self, inputType=None):
        cur = inputType or self.inputType
        out = []
        for i, c in enumerate(self.confs):
            if i in self.inputPreProcessors:
                cur = self.inputPreProcessors[i].getOutputType(cur)
            cur = c.layer.getOutputType(i, cur)
            out.append(cur)
        return out

    def to_json_obj(self):
        return OrderedDict([
            ("backpropType", self.backpropType), ("cacheMode", self.cacheMode),
            ("confs", [c.to_json() for c in self.confs]), ("dataType", self.dataType),
            ("computeDataType", self.computeDataType), ("epochCount", self.epochCount),
            ("inferenceWorkspaceMode", self.inferenceWorkspaceMode),
            ("inputPreProcessors", {str(k): v.to_json() for k, v in self.inputPreProcessors.items()}),
            ("inputType", self.inputType.to_json() if self.inputType else None),
            ("iterationCount", self.iterationCount), ("tbpttBackLength", self.tbpttBackLength),
            ("tbpttFwdLength", self.tbpttFwdLength), ("trainingWorkspaceMode", self.trainingWorkspaceMode),
            ("validateOutputLayerConfig", self.validateOutputLayerConfig)])

    def toJson(self):
        return json.dumps(self.to_json_obj(), indent=2)

    def toYaml(self):
        import yaml
        return yaml.safe_dump(json.loads(self.toJson()), sort_keys=False)

    @staticmethod
    def fromJson(s):
        d = json.loads(s) if isinstance(s, (str, bytes)) else s
        if "confs" not in d:
            raise ValueError("Not a MultiLayerConfiguration JSON (no 'confs')")
        from . import legacy_json
        d = legacy_json.upgrade(d)                # LegacyJsonFormat / BaseNetConfigDeserializer eras -> current layout
        confs = [NeuralNetConfiguration.from_json_obj(c, True) for c in d["confs"]]
        pre = {int(k): preprocessor_from_json(v) for k, v in (d.get("inputPreProcessors") or {}).items()}
        kw = {k: d[k] for k in ("backpropType", "tbpttFwdLength", "tbpttBackLength", "trainingWorkspaceMode",
                                "inferenceWorkspaceMode", "cacheMode", "dataType", "computeDataType",
                                "validateOutputLayerConfig", "iterationCount", "epochCount") if k in d}
        kw["inputType"] = InputType.from_json(d.get("inputType"))
        return MultiLayerConfiguration(confs, pre, **kw)

    @staticmethod
    def fromYaml(s):
        import yaml
        return MultiLayerConfiguration.fromJson(yaml.safe_load(s))

    def __eq__(self, o):
        return isinstance(o, MultiLayerConfiguration) and self.to_json_obj() == o.to_json_obj()


# ---------------------------------------------------------------------------------------------
class GraphBuilder:
    def __init__(self, gb: "NeuralNetConfiguration.Builder"):
        self._gb = gb
        self._vertices: "OrderedDict[str, G.GraphVertex]" = OrderedDict()
        self._vertexInputs: "OrderedDict[str, List[str]]" = OrderedDict()
        self._inputs: List[str] = []
        self._outputs: List[str] = []
        self._inputTypes = None
        self._pre: Dict[str, object] = {}
        self._backpropType = BackpropType.Standard
        self._tbpttFwd = 20
        self._tbpttBack = 20
        self._validateOutput = True
        self._allowDisconnected = False
        self._allowNoOutput = False

    def addInputs(self, *names):
        for n in names:
            if isinstance(n, (list, tuple)):
                self._inputs.extend(n)
            else:
                self._inputs.append(n)
        return self

    def setInputTypes(self, *its):
        self._inputTypes = list(its[0]) if len(its) == 1 and isinstance(its[0], (list, tuple)) else list(its)
        return self

    def addLayer(self, name, layer, *inputs):
        pre = None
        inputs = list(inputs)
        if inputs and not isinstance(inputs[0], str) and inputs[0] is not None and not isinstance(inputs[0], (list, tuple)):
            pre = inputs.pop(0)
        if inputs and isinstance(inputs[0], (list, tuple)):
            inputs = list(inputs[0])
        layer.layerName = name
        self._vertices[name] = ("layer", layer, pre)
        self._vertexInputs[name] = inputs
        return self

    layer = addLayer

    def appendLayer(self, name, layer):
        last = next(reversed(self._vertices)) if self._vertices else self._inputs[-1]
        return self.addLayer(name, layer, last)

    def addVertex(self, name, vertex, *inputs):
        if inputs and isinstance(inputs[0], (list, tuple)):
            inputs = list(inputs[0])
        self._vertices[name] = ("vertex", vertex, None)
        self._vertexInputs[name] = list(inputs)
        return self

    def inputPreProcessor(self, name, p):
        self._pre[name] = p; return self

    def setOutputs(self, *names):
        self._outputs = list(names[0]) if len(names) == 1 and isinstance(names[0], (list, tuple)) else list(names)
        return self

    def removeVertex(self, name, removeConnections=True):
        self._vertices.pop(name, None)
        self._vertexInputs.pop(name, None)
        if removeConnections:
            for k, v in self._vertexInputs.items():
                self._vertexInputs[k] = [x for x in v if x != name]
        if name in self._outputs:
            self._outputs.remove(name)
        return self

    def backpropType(self, t):
        self._backpropType = t; return self

    def tBPTTLength(self, n):
        self._tbpttFwd = self._tbpttBack = int(n); return self

    def tBPTTForwardLength(self, n):
        self._tbpttFwd = int(n); return self

    def tBPTTBackwardLength(self, n):
        self._tbpttBack = int(n); return self

    def validateOutputLayerConfig(self, b):
        self._validateOutput = bool(b); return self

    def allowDisconnected(self, b):
        self._allowDisconnected = bool(b); return self

    def allowNoOutput(self, b):
        self._allowNoOutput = bool(b); return self

    def getVertices(self):
        return self._vertices

    def build(self) -> "ComputationGraphConfiguration":
        verts = OrderedDict()
        for name, (kind, obj, pre) in self._vertices.items():
            if kind == "layer":
                nnc = self._gb._nnc(obj)
                p = self._pre.get(name, pre)
                verts[name] = G.LayerVertex(nnc, p, name in self._outputs)
            else:
                if hasattr(obj, "apply_global"):
                    obj.apply_global(self._gb._g)
                verts[name] = obj
        conf = ComputationGraphConfiguration(verts, OrderedDict((k, list(v)) for k, v in self._vertexInputs.items()),
                                             list(self._inputs), list(self._outputs),
                                             backpropType=self._backpropType, tbpttFwdLength=self._tbpttFwd,
                                             tbpttBackLength=self._tbpttBack, trainingWorkspaceMode=self._gb._trainWs,
                                             inferenceWorkspaceMode=self._gb._infWs, cacheMode=self._gb._cacheMode,
                                             dataType=self._gb._dataType, computeDataType=self._gb._computeDataType,
                                             validateOutputLayerConfig=self._validateOutput,
                                             defaultConfiguration=self._gb._nnc(None))
        conf.validate(self._allowDisconnected, self._allowNoOutput)
        if self._inputTypes is not None:
            conf.inputTypes = list(self._inputTypes)
            conf.addPreProcessors(*self._inputTypes)
        return conf


class ComputationGraphConfiguration:
    def __init__(self, vertices, vertexInputs, networkInputs, networkOutputs, **kw):
        self.vertices: "OrderedDict[str, G.GraphVertex]" = vertices
        self.vertexInputs: "OrderedDict[str, List[str]]" = vertexInputs
        self.networkInputs: List[str] = networkInputs
        self.networkOutputs: List[str] = networkOutputs
        self.backpropType = kw.get("backpropType", BackpropType.Standard)
        self.tbpttFwdLength = kw.get("tbpttFwdLength", 20)
        self.tbpttBackLength = kw.get("tbpttBackLength", 20)
        self.trainingWorkspaceMode = kw.get("trainingWorkspaceMode", WorkspaceMode.ENABLED)
        self.inferenceWorkspaceMode = kw.get("inferenceWorkspaceMode", WorkspaceMode.ENABLED)
        self.cacheMode = kw.get("cacheMode", CacheMode.NONE)
        self.dataType = kw.get("dataType", DataType.FLOAT)
        self.computeDataType = kw.get("computeDataType")
        self.validateOutputLayerConfig = kw.get("validateOutputLayerConfig", True)
        self.defaultConfiguration = kw.get("defaultConfiguration") or NeuralNetConfiguration(None)
        self.iterationCount = kw.get("iterationCount", 0)
        self.epochCount = kw.get("epochCount", 0)
        self.inputTypes = kw.get("inputTypes")
        self.topologicalOrder = None
        self.topologicalOrderStr = None

    def clone(self):
        return copy.deepcopy(self)

    def getVertices(self):
        return self.vertices

    def getVertexInputs(self):
        return self.vertexInputs

    def getNetworkInputs(self):
        return self.networkInputs

    def getNetworkOutputs(self):
        return self.networkOutputs

    def validate(self, allowDisconnected=False, allowNoOutput=False):
        if not self.networkInputs:
            raise ValueError("Invalid configuration: network has no inputs")
        if not self.networkOutputs and not allowNoOutput:
            raise ValueError("Invalid configuration: network has no outputs")
        names = set(self.networkInputs)
        for n in self.vertices:
            if n in names:
                raise ValueError("Invalid configuration: duplicate name '%s'" % n)
            names.add(n)
        for n, ins in self.vertexInputs.items():
            if not ins:
                raise ValueError("Invalid configuration: vertex '%s' has no inputs" % n)
            for i in ins:
                if i not in names:
                    raise ValueError("Invalid configuration: vertex '%s' has input '%s' that does not exist" % (n, i))
            v = self.vertices[n]
            if len(ins) < v.minVertexInputs() or len(ins) > v.maxVertexInputs():
                raise ValueError("Invalid configuration: vertex '%s' (%s) has %d inputs" % (n, type(v).__name__, len(ins)))
        for o in self.networkOutputs:
            if o not in self.vertices:
                raise ValueError("Invalid configuration: output '%s' is not a vertex" % o)
        if not allowDisconnected:
            used = set(self.networkOutputs)
            for ins in self.vertexInputs.values():
                used.update(ins)
            for n in list(self.vertices) + self.networkInputs:
                if n not in used:
                    raise ValueError("Invalid configuration: vertex/input '%s' is disconnected (not used as input to "
                                     "any vertex and not an output)" % n)
        self.topological_order()  # raises on cycles

    def topological_order(self) -> List[str]:
        """Kahn topological sort (ComputationGraph.topologicalSortOrder, ComputationGraph.java:1187)."""
        indeg = {n: 0 for n in self.vertices}
        outs: Dict[str, List[str]] = {n: [] for n in list(self.vertices) + self.networkInputs}
        for n, ins in self.vertexInputs.items():
            for i in ins:
                outs[i].append(n)
                indeg[n] += 1
        order = []
        ready = list(self.networkInputs)
        # deterministic: process in insertion order
        pos = {n: k for k, n in enumerate(list(self.networkInputs) + list(self.vertices))}
        import heapq
        heap = [(pos[n], n) for n in ready]
        heapq.heapify(heap)
        while heap:
            _, n = heapq.heappop(heap)
            order.append(n)
            for m in outs[n]:
                indeg[m] -= 1
                if indeg[m] == 0:
                    heapq.heappush(heap, (pos[m], m))
        if len(order) != len(self.vertices) + len(self.networkInputs):
            raise ValueError("Invalid configuration: cycle detected in graph")
        return order

    def getMemoryReport(self, *inputTypes):
        from .memory import memory_report_graph
        return memory_report_graph(self, *inputTypes)

    def getLayerActivationTypes(self, *inputTypes):
        its = list(inputTypes) if inputTypes else list(self.inputTypes or [])
        types = dict(zip(self.networkInputs, its))
        for idx, n in enumerate(self.topological_order()):
            if n in types:
                continue
            v = self.vertices[n]
            types[n] = v.getOutputType(idx, *[types[i] for i in self.vertexInputs[n]])
        return types

    def addPreProcessors(self, *inputTypes):
        """Infer nIn for every layer and insert InputPreProcessors (ComputationGraphConfiguration.addPreProcessors)."""
        types = dict(zip(self.networkInputs, inputTypes))
        for idx, n in enumerate(self.topological_order()):
            if n in types:
                continue
            v = self.vertices[n]
            ins = [types[i] for i in self.vertexInputs[n]]
            if isinstance(v, G.LayerVertex):
                l = v.layer
                it = ins[0]
                if v.preProcessor is None:
                    v.preProcessor = l.getPreProcessorForInputType(it)
                if v.preProcessor is not None:
                    it = v.preProcessor.getOutputType(it)
                l.setNIn(it, True)
                types[n] = l.getOutputType(idx, it)
            else:
                if isinstance(v, G.AttentionVertex) and v.projectInput and not v.nInQueries:
                    v.nInQueries, v.nInKeys, v.nInValues = ins[0].size, ins[1].size, ins[2].size
                types[n] = v.getOutputType(idx, *ins)
        return types

    def to_json_obj(self):
        return OrderedDict([
            ("backpropType", self.backpropType), ("cacheMode", self.cacheMode), ("dataType", self.dataType),
            ("computeDataType", self.computeDataType),
            ("defaultConfiguration", self.defaultConfiguration.to_json()), ("epochCount", self.epochCount),
            ("inferenceWorkspaceMode", self.inferenceWorkspaceMode),
            ("inputTypes", [t.to_json() for t in self.inputTypes] if self.inputTypes else None),
            ("iterationCount", self.iterationCount),
            ("networkInputs", self.networkInputs), ("networkOutputs", self.networkOutputs),
            ("tbpttBackLength", self.tbpttBackLength), ("tbpttFwdLength", self.tbpttFwdLength),
            ("topologicalOrder", None), ("topologicalOrderStr", None),
            ("trainingWorkspaceMode", self.trainingWorkspaceMode),
            ("validateOutputLayerConfig", self.validateOutputLayerConfig),
            ("vertexInputs", OrderedDict(self.vertexInputs)),
            ("vertices", OrderedDict((k, v.to_json()) for k, v in self.vertices.items()))])

    def toJson(self):
        return json.dumps(self.to_json_obj(), indent=2)

    def toYaml(self):
        import yaml
        return yaml.safe_dump(json.loads(self.toJson()), sort_keys=False)

    @staticmethod
    def fromJson(s):
        d = json.loads(s, object_pairs_hook=OrderedDict) if isinstance(s, (str, bytes)) else s
        if "vertices" not in d:
            raise ValueError("Not a ComputationGraphConfiguration JSON (no 'vertices')")
        from . import legacy_json
        d = legacy_json.upgrade(d)
        verts = OrderedDict((k, G.vertex_from_json(v)) for k, v in d["vertices"].items())
        vin = OrderedDict((k, list(v)) for k, v in d["vertexInputs"].items())
        # keep vertices in vertexInputs order when present (stable param layout)
        kw = {k: d[k] for k in ("backpropType", "tbpttFwdLength", "tbpttBackLength", "trainingWorkspaceMode",
                                "inferenceWorkspaceMode", "cacheMode", "dataType", "computeDataType",
                                "validateOutputLayerConfig", "iterationCount", "epochCount") if k in d}
        if d.get("defaultConfiguration"):
            kw["defaultConfiguration"] = NeuralNetConfiguration.from_json_obj(d["defaultConfiguration"])
        if d.get("inputTypes"):
            kw["inputTypes"] = [InputType.from_json(t) for t in d["inputTypes"]]
        return ComputationGraphConfiguration(verts, vin, list(d["networkInputs"]), list(d["networkOutputs"]), **kw)

    @staticmethod
    def fromYaml(s):
        import yaml
        return ComputationGraphConfiguration.fromJson(yaml.safe_load(s))

    def __eq__(self, o):
        return isinstance(o, ComputationGraphConfiguration) and self.to_json_obj() == o.to_json_obj()
