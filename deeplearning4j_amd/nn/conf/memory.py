"""Memory estimates for a network configuration.

Reference: deeplearning4j-nn nn/conf/memory/{MemoryReport,LayerMemoryReport,NetworkMemoryReport,MemoryType,MemoryUseMode}.java
and ``MultiLayerConfiguration.getMemoryReport(InputType)`` / ``ComputationGraphConfiguration.getMemoryReport(InputType...)``.
Here the estimate follows what THIS runtime actually allocates on an MI355X: one flat fp32 parameter vector + one flat
gradient vector + flat updater state, bf16 compute replicas of the weights when computeDataType=BFLOAT16, and one
activation (plus its gradient during training) per layer, sized from the InputType shape inference.
"""
from __future__ import annotations

import json
from collections import OrderedDict



class MemoryType:
    PARAMETERS, PARAMATER_GRADIENTS, ACTIVATIONS, ACTIVATION_GRADIENTS, UPDATER_STATE, WORKING_MEMORY_FIXED, WORKING_MEMORY_VARIABLE, \
        CACHED_MEMORY_FIXED, CACHED_MEMORY_VARIABLE = ("PARAMETERS", "PARAMATER_GRADIENTS", "ACTIVATIONS", "ACTIVATION_GRADIENTS",
                                                       "UPDATER_STATE", "WORKING_MEMORY_FIXED", "WORKING_MEMORY_VARIABLE",
                                                       "CACHED_MEMORY_FIXED", "CACHED_MEMORY_VARIABLE")
    ALL = [PARAMETERS, PARAMATER_GRADIENTS, ACTIVATIONS, ACTIVATION_GRADIENTS, UPDATER_STATE, WORKING_MEMORY_FIXED,
           WORKING_MEMORY_VARIABLE, CACHED_MEMORY_FIXED, CACHED_MEMORY_VARIABLE]
    INFERENCE = [PARAMETERS, ACTIVATIONS, WORKING_MEMORY_FIXED, WORKING_MEMORY_VARIABLE]


class MemoryUseMode:
    INFERENCE, TRAINING = "INFERENCE", "TRAINING"


def _bytes_of(dtype_name):
    return {"DOUBLE": 8, "FLOAT": 4, "HALF": 2, "BFLOAT16": 2}.get(str(dtype_name).upper(), 4)


class LayerMemoryReport:
    def __init__(self, layerName, layerType, inputType, outputType, numParams, updaterStateSize, workingElemsPerEx=0):
        self.layerName, self.layerType, self.inputType, self.outputType = layerName, layerType, inputType, outputType
        self.parameterSize, self.updaterStateSize = int(numParams), int(updaterStateSize)
        self.activationElemsPerEx = int(outputType.arrayElementsPerExample()) if outputType is not None else 0
        self.workingElemsPerEx = int(workingElemsPerEx)

    def getName(self):
        return self.layerName

    def getMemoryBytes(self, memoryType, minibatchSize, useMode, cacheMode=None, dataType="FLOAT", computeDataType=None):
        pb = _bytes_of(dataType)
        ab = _bytes_of(computeDataType or dataType)
        train = useMode == MemoryUseMode.TRAINING
        if memoryType == MemoryType.PARAMETERS:
            extra = self.parameterSize * ab if (computeDataType and _bytes_of(computeDataType) != pb) else 0   # bf16 replicas
            return self.parameterSize * pb + extra
        if memoryType == MemoryType.PARAMATER_GRADIENTS:
            return self.parameterSize * pb if train else 0
        if memoryType == MemoryType.UPDATER_STATE:
            return self.updaterStateSize * pb if train else 0
        if memoryType == MemoryType.ACTIVATIONS:
            return self.activationElemsPerEx * minibatchSize * ab
        if memoryType == MemoryType.ACTIVATION_GRADIENTS:
            return (int(self.inputType.arrayElementsPerExample()) * minibatchSize * ab) if (train and self.inputType is not None) else 0
        if memoryType == MemoryType.WORKING_MEMORY_VARIABLE:
            return self.workingElemsPerEx * minibatchSize * ab
        return 0

    def getTotalMemoryBytes(self, minibatchSize, useMode, cacheMode=None, dataType="FLOAT", computeDataType=None):
        types = MemoryType.ALL if useMode == MemoryUseMode.TRAINING else MemoryType.INFERENCE
        return sum(self.getMemoryBytes(t, minibatchSize, useMode, cacheMode, dataType, computeDataType) for t in types)

    def to_json(self):
        return OrderedDict([("layerName", self.layerName), ("layerType", self.layerType), ("parameterSize", self.parameterSize),
                            ("updaterStateSize", self.updaterStateSize), ("activationElemsPerEx", self.activationElemsPerEx),
                            ("workingElemsPerEx", self.workingElemsPerEx)])


class NetworkMemoryReport:
    def __init__(self, layerReports, modelClass, modelName, inputTypes, dataType="FLOAT", computeDataType=None):
        self.layerAndVertexReports = OrderedDict((r.getName(), r) for r in layerReports)
        self.modelClass, self.modelName, self.networkInputTypes = modelClass, modelName, list(inputTypes)
        self.dataType, self.computeDataType = dataType, computeDataType

    def getMemoryBytes(self, memoryType, minibatchSize, useMode, cacheMode=None):
        return sum(r.getMemoryBytes(memoryType, minibatchSize, useMode, cacheMode, self.dataType, self.computeDataType)
                   for r in self.layerAndVertexReports.values())

    def getTotalMemoryBytes(self, minibatchSize, useMode, cacheMode=None):
        return sum(r.getTotalMemoryBytes(minibatchSize, useMode, cacheMode, self.dataType, self.computeDataType)
                   for r in self.layerAndVertexReports.values())

    def toString(self, minibatchSize=32):
        lines = ["----- Network Memory Report -----", "  Model Class: %s" % self.modelClass, "  Model Name: %s" % self.modelName,
                 "  Network Input: %s" % [str(t) for t in self.networkInputTypes], "  # Layers: %d" % len(self.layerAndVertexReports),
                 "  Data type: %s%s" % (self.dataType, " (compute %s)" % self.computeDataType if self.computeDataType else "")]
        for mode in (MemoryUseMode.INFERENCE, MemoryUseMode.TRAINING):
            tot = self.getTotalMemoryBytes(minibatchSize, mode)
            lines.append("  %s, minibatch %d: %d bytes (%.2f MiB)" % (mode, minibatchSize, tot, tot / 2 ** 20))
            for t in MemoryType.ALL:
                b = self.getMemoryBytes(t, minibatchSize, mode)
                if b:
                    lines.append("      %-26s %14d" % (t, b))
        return "\n".join(lines)

    __str__ = toString

    def toJson(self):
        return json.dumps({"modelClass": self.modelClass, "modelName": self.modelName, "dataType": self.dataType,
                           "computeDataType": self.computeDataType,
                           "layers": [r.to_json() for r in self.layerAndVertexReports.values()]}, indent=2)


def _layer_report(name, layer, in_type, out_type):
    from ...nd4j.learning import NoOp
    state = 0
    try:
        n_params = int(layer.numParams())
    except Exception:
        n_params = 0
    try:
        for spec in layer.param_specs():
            upd = layer.getUpdaterByParam(spec)
            if upd is not None and not isinstance(upd, NoOp) and hasattr(upd, "stateSize"):
                state += int(upd.stateSize(spec.numel))
    except Exception:
        state = 0
    return LayerMemoryReport(name, type(layer).__name__, in_type, out_type, n_params, state)


def memory_report_mln(conf, inputType=None):
    """MultiLayerConfiguration.getMemoryReport (reference MultiLayerConfiguration.java getMemoryReport)."""
    cur = inputType or conf.inputType
    if cur is None:
        raise ValueError("getMemoryReport needs an InputType")
    reps = []
    for i, c in enumerate(conf.confs):
        if i in conf.inputPreProcessors:
            cur = conf.inputPreProcessors[i].getOutputType(cur)
        out = c.layer.getOutputType(i, cur)
        reps.append(_layer_report(getattr(c.layer, "layerName", None) or str(i), c.layer, cur, out))
        cur = out
    return NetworkMemoryReport(reps, "MultiLayerNetwork", "MultiLayerNetwork", [inputType or conf.inputType],
                               str(conf.dataType),
                               conf.computeDataType)


def memory_report_graph(conf, *inputTypes):
    from . import graph as G
    its = list(inputTypes) if inputTypes else list(conf.inputTypes or [])
    types = dict(zip(conf.networkInputs, its))
    reps = []
    for idx, n in enumerate(conf.topological_order()):
        if n in types:
            continue
        v = conf.vertices[n]
        ins = [types[i] for i in conf.vertexInputs[n]]
        if isinstance(v, G.LayerVertex):
            it = ins[0]
            if v.preProcessor is not None:
                it = v.preProcessor.getOutputType(it)
            out = v.layer.getOutputType(idx, it)
            reps.append(_layer_report(n, v.layer, it, out))
        else:
            out = v.getOutputType(idx, *ins)
            reps.append(LayerMemoryReport(n, type(v).__name__, ins[0] if ins else None, out, 0, 0))
        types[n] = out
    return NetworkMemoryReport(reps, "ComputationGraph", "ComputationGraph", its,
                               str(conf.dataType),
                               conf.computeDataType)


class MemoryReport:
    """Common base name of the reports (nn/conf/memory/MemoryReport.java): static helpers shared by LayerMemoryReport / NetworkMemoryReport."""

    @staticmethod
    def fromLayer(conf, inputType=None):
        return conf if isinstance(conf, LayerMemoryReport) else conf.getMemoryReport(inputType)

    @staticmethod
    def isReport(obj):
        return isinstance(obj, (LayerMemoryReport, NetworkMemoryReport))
