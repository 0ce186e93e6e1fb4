"""Transfer learning (reference: deeplearning4j-nn/.../nn/transferlearning/{TransferLearning,FineTuneConfiguration,
TransferLearningHelper}.java).

The builders edit a *copy* of the source network's configuration (freeze / replace nOut / remove / append), construct a new
network, and copy every retained parameter tensor across by name; replaced layers (and the layer that consumes a replaced
nOut) are re-initialised with the requested weight init, as in the reference (TransferLearning.java:230-330, 560-780).
"""
from __future__ import annotations

import copy
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from ..datasets.dataset import DataSet
from .conf import graph as G
from .conf.base import get_weight_init
from .conf.configuration import ComputationGraphConfiguration, MultiLayerConfiguration, NeuralNetConfiguration
from .conf.layers import FrozenLayer, INHERITABLE, Layer, _ALIASES
from ..nd4j.activations import get_activation


class FineTuneConfiguration:
    """Overrides applied to every non-frozen layer of the source network and used as defaults for appended layers."""
    _GLOBAL = ("seed", "miniBatch", "minimize", "optimizationAlgo", "maxNumLineSearchIterations", "trainingWorkspaceMode",
               "inferenceWorkspaceMode", "backpropType", "tbpttFwdLength", "tbpttBackLength", "cacheMode", "dataType", "computeDataType")

    def __init__(self, **kw):
        self.layer: Dict[str, object] = {}
        self.glob: Dict[str, object] = {}
        for k, v in kw.items():
            getattr(self.Builder._of(self), k)(v)

    class Builder:
        def __init__(self):
            self._c = FineTuneConfiguration()

        @classmethod
        def _of(cls, c):
            b = cls.__new__(cls)
            b._c = c
            return b

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)

            def setter(*v):
                val = v[0] if len(v) == 1 else list(v)
                if name in FineTuneConfiguration._GLOBAL:
                    self._c.glob[name] = val
                elif name == "tBPTTLength":
                    self._c.glob["tbpttFwdLength"] = self._c.glob["tbpttBackLength"] = int(val)
                else:
                    f = _ALIASES.get(name, name)
                    if f not in INHERITABLE:
                        raise AttributeError("FineTuneConfiguration has no option '%s'" % name)
                    if f == "activationFn":
                        val = get_activation(val)
                    elif f == "weightInitFn":
                        val = get_weight_init(val)
                    self._c.layer[f] = val
                return self
            return setter

        def build(self):
            return self._c

    @staticmethod
    def builder():
        return FineTuneConfiguration.Builder()

    # -- application
    def applyToLayer(self, l: Layer):
        target = l.underlying if isinstance(l, FrozenLayer) else l
        for f, v in self.layer.items():
            if hasattr(target, f):
                setattr(target, f, copy.deepcopy(v))
        target._normalize()

    def applyToNNC(self, nnc: NeuralNetConfiguration):
        for k in ("seed", "miniBatch", "minimize", "optimizationAlgo", "maxNumLineSearchIterations", "cacheMode", "dataType"):
            if k in self.glob:
                setattr(nnc, k, self.glob[k])
        if nnc.layer is not None:
            self.applyToLayer(nnc.layer)

    def applyToConf(self, conf):
        for k in ("trainingWorkspaceMode", "inferenceWorkspaceMode", "backpropType", "tbpttFwdLength", "tbpttBackLength", "cacheMode",
                  "dataType", "computeDataType"):
            if k in self.glob:
                setattr(conf, k, self.glob[k])

    def defaults_for_new_layer(self, template: Optional[NeuralNetConfiguration]) -> Dict[str, object]:
        g = dict(NeuralNetConfiguration.Builder()._g)
        g.update(self.layer)
        return g

    def toJson(self):
        import json
        ser = lambda v: v.to_json() if hasattr(v, "to_json") else v
        return json.dumps({"layer": {k: ser(v) for k, v in self.layer.items()}, "global": {k: ser(v) for k, v in self.glob.items()}},
                          default=str, sort_keys=True)


def _copy_params(src_table: Dict[str, torch.Tensor], dst_net, key_map: Dict[str, str], skip: set):
    dst = dst_net.paramTable()
    for dk, t in dst.items():
        layer_key, _, pkey = dk.rpartition("_")
        sk = key_map.get(layer_key)
        if sk is None or layer_key in skip:
            continue
        s = src_table.get("%s_%s" % (sk, pkey))
        if s is not None and tuple(s.shape) == tuple(t.shape):
            t.copy_(s.to(t.dtype))
    dst_net._refresh_replicas()


class TransferLearning:
    # ------------------------------------------------------------------ MultiLayerNetwork
    class Builder:
        def __init__(self, origModel):
            from .multilayer import MultiLayerNetwork
            if not isinstance(origModel, MultiLayerNetwork):
                raise TypeError("TransferLearning.Builder needs a MultiLayerNetwork (use GraphBuilder for a ComputationGraph)")
            self._orig = origModel
            self._conf: MultiLayerConfiguration = origModel.conf.clone()
            self._ftc: Optional[FineTuneConfiguration] = None
            self._frozenTill = -1
            self._src_index: List[Optional[int]] = list(range(len(self._conf.confs)))   # new layer i <- source layer
            self._reinit: set = set()
            self._inputType = self._conf.inputType
            self._pre = dict(self._conf.inputPreProcessors)
            self._popped = False

        def fineTuneConfiguration(self, ftc):
            self._ftc = ftc
            return self

        def setFeatureExtractor(self, layerNum):
            self._frozenTill = int(layerNum)
            return self

        def nOutReplace(self, layerNum, nOut, scheme="XAVIER", schemeNext=None):
            confs = self._conf.confs
            l = confs[layerNum].layer
            tgt = l.underlying if isinstance(l, FrozenLayer) else l
            tgt.nOut = int(nOut)
            tgt.weightInitFn = get_weight_init(scheme)
            self._reinit.add(layerNum)
            if layerNum + 1 < len(confs):
                nxt = confs[layerNum + 1].layer
                nt = nxt.underlying if isinstance(nxt, FrozenLayer) else nxt
                if hasattr(nt, "nIn"):
                    nt.nIn = int(nOut)
                    nt.weightInitFn = get_weight_init(schemeNext if schemeNext is not None else scheme)
                    self._reinit.add(layerNum + 1)
            return self

        def nInReplace(self, layerNum, nIn, scheme="XAVIER"):
            l = self._conf.confs[layerNum].layer
            tgt = l.underlying if isinstance(l, FrozenLayer) else l
            tgt.nIn = int(nIn)
            tgt.weightInitFn = get_weight_init(scheme)
            self._reinit.add(layerNum)
            return self

        def removeOutputLayer(self):
            return self.removeLayersFromOutput(1)

        def removeLayersFromOutput(self, n):
            for _ in range(int(n)):
                idx = len(self._conf.confs) - 1
                self._conf.confs.pop()
                self._src_index.pop()
                self._pre.pop(idx, None)
                self._reinit.discard(idx)
            self._popped = True
            return self

        def addLayer(self, layer: Layer):
            g = (self._ftc or FineTuneConfiguration()).defaults_for_new_layer(None)
            layer = layer.clone()
            layer.apply_global(g)
            template = self._conf.confs[0] if self._conf.confs else NeuralNetConfiguration(None)
            nnc = template.clone()
            nnc.layer = layer
            prev = self._conf.confs[-1].layer if self._conf.confs else None
            if prev is not None and getattr(layer, "nIn", None) in (0, None) and hasattr(layer, "nIn") and self._inputType is None:
                pt = prev.underlying if isinstance(prev, FrozenLayer) else prev
                layer.nIn = getattr(pt, "nOut", 0)
            self._conf.confs.append(nnc)
            self._src_index.append(None)
            return self

        def setInputPreProcessor(self, layer, processor):
            self._pre[int(layer)] = processor
            return self

        def setInputType(self, it):
            self._inputType = it
            return self

        def validateOutputLayerConfig(self, b):
            self._conf.validateOutputLayerConfig = bool(b)
            return self

        def build(self):
            from .multilayer import MultiLayerNetwork
            conf = self._conf
            for i, nnc in enumerate(conf.confs):
                if self._ftc is not None and i > self._frozenTill:
                    self._ftc.applyToNNC(nnc)
                if i <= self._frozenTill and not isinstance(nnc.layer, FrozenLayer):
                    nnc.layer = FrozenLayer(nnc.layer)
            if self._ftc is not None:
                self._ftc.applyToConf(conf)
            conf.inputPreProcessors = {k: v for k, v in self._pre.items() if k < len(conf.confs)}
            conf.inputType = self._inputType
            if self._inputType is not None:
                conf._apply_input_type(self._inputType, True)
            net = MultiLayerNetwork(conf, device=self._orig.ctx.device)
            net.init()
            key_map = {str(i): str(s) for i, s in enumerate(self._src_index) if s is not None}
            _copy_params(self._orig.paramTable(), net, key_map, {str(i) for i in self._reinit})
            return net

    # ------------------------------------------------------------------ ComputationGraph
    class GraphBuilder:
        def __init__(self, origGraph):
            from .graph.computation_graph import ComputationGraph
            if not isinstance(origGraph, ComputationGraph):
                raise TypeError("TransferLearning.GraphBuilder needs a ComputationGraph")
            self._orig = origGraph
            self._conf: ComputationGraphConfiguration = origGraph.conf.clone()
            self._ftc: Optional[FineTuneConfiguration] = None
            self._frozen_outputs: List[str] = []
            self._reinit: set = set()
            self._new: set = set()
            self._inputTypes = list(self._conf.inputTypes) if self._conf.inputTypes else None

        def fineTuneConfiguration(self, ftc):
            self._ftc = ftc
            return self

        def setFeatureExtractor(self, *names):
            self._frozen_outputs = list(names[0]) if len(names) == 1 and isinstance(names[0], (list, tuple)) else list(names)
            return self

        def _layer_of(self, name):
            v = self._conf.vertices[name]
            if not isinstance(v, G.LayerVertex):
                raise ValueError("vertex '%s' is not a layer" % name)
            l = v.layer
            return l.underlying if isinstance(l, FrozenLayer) else l

        def nOutReplace(self, layerName, nOut, scheme="XAVIER", schemeNext=None):
            l = self._layer_of(layerName)
            l.nOut = int(nOut)
            l.weightInitFn = get_weight_init(scheme)
            self._reinit.add(layerName)
            # every layer fed (possibly through param-less vertices) by this one gets a new nIn and is re-initialised
            frontier, seen = [layerName], set()
            while frontier:
                cur = frontier.pop()
                for n, ins in self._conf.vertexInputs.items():
                    if cur in ins and n not in seen:
                        seen.add(n)
                        v = self._conf.vertices[n]
                        if isinstance(v, G.LayerVertex) and v.layer.param_specs():
                            t = self._layer_of(n)
                            if hasattr(t, "nIn") and len(ins) == 1 and cur == layerName:
                                t.nIn = int(nOut)
                            t.weightInitFn = get_weight_init(schemeNext if schemeNext is not None else scheme)
                            self._reinit.add(n)
                        else:
                            frontier.append(n)
            return self

        def nInReplace(self, layerName, nIn, scheme="XAVIER"):
            l = self._layer_of(layerName)
            l.nIn = int(nIn)
            l.weightInitFn = get_weight_init(scheme)
            self._reinit.add(layerName)
            return self

        def removeVertexKeepConnections(self, name):
            self._conf.vertices.pop(name)
            self._conf.vertexInputs.pop(name)
            if name in self._conf.networkOutputs:
                self._conf.networkOutputs.remove(name)
            self._removed_keep = getattr(self, "_removed_keep", set()) | {name}
            return self

        def removeVertexAndConnections(self, name):
            self._conf.vertices.pop(name)
            self._conf.vertexInputs.pop(name)
            for k, ins in self._conf.vertexInputs.items():
                self._conf.vertexInputs[k] = [i for i in ins if i != name]
            if name in self._conf.networkOutputs:
                self._conf.networkOutputs.remove(name)
            return self

        def addLayer(self, name, layer: Layer, *inputs):
            pre = None
            inputs = list(inputs)
            if inputs and not isinstance(inputs[0], str):
                pre = inputs.pop(0)
            g = (self._ftc or FineTuneConfiguration()).defaults_for_new_layer(None)
            layer = layer.clone()
            layer.layerName = name
            layer.apply_global(g)
            nnc = self._conf.defaultConfiguration.clone()
            nnc.layer = layer
            self._conf.vertices[name] = G.LayerVertex(nnc, pre, False)
            self._conf.vertexInputs[name] = inputs
            self._new.add(name)
            return self

        def addVertex(self, name, vertex, *inputs):
            self._conf.vertices[name] = vertex
            self._conf.vertexInputs[name] = list(inputs)
            self._new.add(name)
            return self

        def addInputs(self, *names):
            self._conf.networkInputs += list(names)
            return self

        def setOutputs(self, *names):
            self._conf.networkOutputs = list(names)
            return self

        def setInputTypes(self, *its):
            self._inputTypes = list(its)
            return self

        def _frozen_set(self):
            """All vertices on a path from the inputs to (and including) the feature-extractor vertices."""
            frozen, stack = set(), list(self._frozen_outputs)
            while stack:
                n = stack.pop()
                if n in frozen or n in self._conf.networkInputs:
                    continue
                frozen.add(n)
                stack += self._conf.vertexInputs.get(n, [])
            return frozen

        def build(self):
            from .graph.computation_graph import ComputationGraph
            conf = self._conf
            frozen = self._frozen_set()
            for n, v in conf.vertices.items():
                if isinstance(v, G.LayerVertex):
                    if n in frozen:
                        if not isinstance(v.layer, FrozenLayer):
                            v.layerConf.layer = FrozenLayer(v.layer)
                    elif self._ftc is not None and n not in self._new:
                        self._ftc.applyToNNC(v.layerConf)
                    v.outputVertex = n in conf.networkOutputs
                elif n in frozen and not isinstance(v, G.FrozenVertex) and False:
                    pass
            if self._ftc is not None:
                self._ftc.applyToConf(conf)
                self._ftc.applyToNNC(conf.defaultConfiguration)
            conf.validate(False, False)
            if self._inputTypes is not None:
                conf.inputTypes = list(self._inputTypes)
                conf.addPreProcessors(*self._inputTypes)
            net = ComputationGraph(conf, device=self._orig.ctx.device)
            net.init()
            key_map = {n: n for n in conf.vertices if n not in self._new}
            _copy_params(self._orig.paramTable(), net, key_map, set(self._reinit))
            return net


class TransferLearningHelper:
    """Featurise once through the frozen part, then train only the unfrozen tail (TransferLearningHelper.java)."""

    def __init__(self, net, frozenTill=None):
        from .multilayer import MultiLayerNetwork
        self.net = net
        self.isGraph = not isinstance(net, MultiLayerNetwork)
        if self.isGraph:
            self._init_graph(frozenTill)
        else:
            self._init_mln(frozenTill)

    # -- MultiLayerNetwork
    def _init_mln(self, frozenTill):
        from .multilayer import MultiLayerNetwork
        net = self.net
        if frozenTill is None:
            frozenTill = -1
            for i, c in enumerate(net.conf.confs):
                if isinstance(c.layer, FrozenLayer):
                    frozenTill = i
        else:
            for i in range(frozenTill + 1):
                c = net.conf.confs[i]
                if not isinstance(c.layer, FrozenLayer):
                    c.layer = FrozenLayer(c.layer)
        if frozenTill < 0:
            raise ValueError("No frozen layers: nothing to featurize")
        self.frozenTill = frozenTill
        conf = net.conf
        tail = MultiLayerConfiguration([c.clone() for c in conf.confs[frozenTill + 1:]],
                                       {k - frozenTill - 1: v for k, v in conf.inputPreProcessors.items() if k > frozenTill},
                                       backpropType=conf.backpropType, tbpttFwdLength=conf.tbpttFwdLength, tbpttBackLength=conf.tbpttBackLength,
                                       dataType=conf.dataType, computeDataType=conf.computeDataType,
                                       validateOutputLayerConfig=conf.validateOutputLayerConfig)
        self._tail = MultiLayerNetwork(tail, device=net.ctx.device)
        self._tail.init()
        self._map = {str(i - frozenTill - 1): str(i) for i in range(frozenTill + 1, len(conf.confs))}
        _copy_params(net.paramTable(), self._tail, self._map, set())

    def _sync_back(self):
        src, dst = self._tail.paramTable(), self.net.paramTable()
        for tk, t in src.items():
            lk, _, pk = tk.rpartition("_")
            d = dst.get("%s_%s" % (self._map[lk], pk))
            if d is not None:
                d.copy_(t)
        self.net._refresh_replicas()

    # -- ComputationGraph
    def _init_graph(self, frozenOutputs):
        from .graph.computation_graph import ComputationGraph
        net = self.net
        conf = net.conf
        if frozenOutputs is None:
            frozen = {n for n, v in conf.vertices.items() if isinstance(v, G.LayerVertex) and isinstance(v.layer, FrozenLayer)}
        else:
            names = [frozenOutputs] if isinstance(frozenOutputs, str) else list(frozenOutputs)
            frozen, stack = set(), names
            while stack:
                n = stack.pop()
                if n in frozen or n in conf.networkInputs:
                    continue
                frozen.add(n)
                stack += conf.vertexInputs.get(n, [])
        if not frozen:
            raise ValueError("No frozen layers: nothing to featurize")
        # param-less vertices whose inputs are all frozen belong to the frozen part too
        changed = True
        while changed:
            changed = False
            for n, ins in conf.vertexInputs.items():
                if n not in frozen and not (isinstance(conf.vertices[n], G.LayerVertex) and conf.vertices[n].layer.param_specs()) \
                        and n not in conf.networkOutputs and all(i in frozen for i in ins):
                    frozen.add(n)
                    changed = True
        boundary = [n for n in conf.topological_order() if n in frozen and
                    any(n in ins and m not in frozen for m, ins in conf.vertexInputs.items())]
        for i in conf.networkInputs:
            if any(i in ins and m not in frozen for m, ins in conf.vertexInputs.items()):
                raise ValueError("Unfrozen vertex reads network input '%s' directly; featurization would drop it" % i)
        self.frozenInputVertices = boundary
        verts = OrderedDict((n, copy.deepcopy(v)) for n, v in conf.vertices.items() if n not in frozen)
        vin = OrderedDict((n, list(conf.vertexInputs[n])) for n in verts)
        tail = ComputationGraphConfiguration(verts, vin, list(boundary), list(conf.networkOutputs), backpropType=conf.backpropType,
                                             tbpttFwdLength=conf.tbpttFwdLength, tbpttBackLength=conf.tbpttBackLength,
                                             dataType=conf.dataType, computeDataType=conf.computeDataType,
                                             defaultConfiguration=conf.defaultConfiguration.clone())
        tail.validate(True, False)
        self._tail = ComputationGraph(tail, device=net.ctx.device)
        self._tail.init()
        self._map = {n: n for n in verts}
        _copy_params(net.paramTable(), self._tail, self._map, set())

    # -- API
    def unfrozenMLN(self):
        return self._tail

    unfrozenGraph = unfrozenMLN

    def featurize(self, ds):
        with torch.no_grad():
            if not self.isGraph:
                f = self.net.feedForwardToLayer(self.frozenTill, ds.features, train=False)[-1]
                f = f.tensor() if hasattr(f, "tensor") and callable(f.tensor) else f
                return DataSet(f, ds.labels, None, ds.labelsMask)
            xs = ds.features if isinstance(ds.features, (list, tuple)) else [ds.features]
            acts = self.net.feedForward(*xs, train=False)
            feats = [acts[n] for n in self.frozenInputVertices]
            feats = [f.tensor() if hasattr(f, "tensor") and callable(f.tensor) else f for f in feats]
            from ..datasets.dataset import MultiDataSet
            labels = ds.labels if isinstance(ds.labels, (list, tuple)) else [ds.labels]
            return MultiDataSet(feats, list(labels))

    def fitFeaturized(self, data):
        if hasattr(data, "hasNext") or (hasattr(data, "__iter__") and not isinstance(data, DataSet) and not hasattr(data, "features")):
            for ds in data:
                self._tail.fit(ds)
        else:
            self._tail.fit(data)
        self._sync_back()

    def outputFromFeaturized(self, features):
        if self.isGraph:
            xs = features if isinstance(features, (list, tuple)) else [features]
            return self._tail.output(*xs)
        return self._tail.output(features)
