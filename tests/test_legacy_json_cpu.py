"""Legacy configuration JSON (wrapper-object subtypes of <= 1.0.0-alpha, enum-era fields of the 0.x line) is upgraded to the current
format on read. Old-format documents are hand-written from the subtype names of LegacyJsonFormat.java:40-187 and the field names
handled by BaseNetConfigDeserializer.java:120-321 (the reference tree ships no regression JSON: parity with real old files is unpinned)."""
import json

import pytest
import torch

from deeplearning4j_amd.nn.conf import legacy_json as LJ
from deeplearning4j_amd.nn.conf.configuration import ComputationGraphConfiguration, MultiLayerConfiguration, NeuralNetConfiguration
from deeplearning4j_amd.nn.conf.layers import DenseLayer, OutputLayer, ConvolutionLayer, SubsamplingLayer, LSTM, RnnOutputLayer
from deeplearning4j_amd.nd4j import learning as U


def _old_050_mln():
    """0.5.0 style: wrapper layers, activationFunction / lossFunction / updater as strings, scalar l1 / l2, weightInit enum, dropOut."""
    def layer(kind, body):
        base = {"activationFunction": "relu", "weightInit": "XAVIER", "updater": "NESTEROVS", "learningRate": 0.05, "momentum": 0.8,
                "rho": 0.0, "rmsDecay": 0.0, "adamMeanDecay": 0.0, "adamVarDecay": 0.0, "l1": 1e-5, "l2": 3e-4, "dropOut": 0.5,
                "biasInit": 0.1, "layerName": None}
        base.update(body)
        return {"layer": {kind: base}, "seed": 42, "miniBatch": True, "minimize": True, "numIterations": 1, "useRegularization": True,
                "optimizationAlgo": "STOCHASTIC_GRADIENT_DESCENT", "variables": ["W", "b"]}
    return {"backprop": True, "pretrain": False, "backpropType": "Standard", "tbpttFwdLength": 20, "tbpttBackLength": 20,
            "confs": [layer("dense", {"nIn": 4, "nOut": 6}),
                      layer("output", {"nIn": 6, "nOut": 3, "activationFunction": "softmax", "lossFunction": "MCXENT", "dropOut": 0.0})],
            "inputPreProcessors": {}}


def test_050_era_multilayer_json_upgrades_and_builds_a_network():
    old = _old_050_mln()
    assert LJ.is_legacy(old)
    conf = MultiLayerConfiguration.fromJson(json.dumps(old))
    l0, l1 = conf.confs[0].layer, conf.confs[1].layer
    assert isinstance(l0, DenseLayer) and isinstance(l1, OutputLayer) and (l0.nIn, l0.nOut, l1.nIn, l1.nOut) == (4, 6, 6, 3)
    assert type(l0.activationFn).__name__ == "ActivationReLU" and type(l1.activationFn).__name__ == "ActivationSoftmax"
    assert type(l1.lossFn).__name__ == "LossMCXENT"
    assert isinstance(l0.iupdater, U.Nesterovs) and l0.iupdater.learningRate == 0.05 and l0.iupdater.momentum == 0.8
    assert l0.weightInitFn.scheme == "XAVIER" and l0.biasInit == 0.1
    assert l0.l1 == 1e-5 and l0.idropout is not None and l0.idropout.p == 0.5 and l1.idropout is None
    # l2 arrives as WeightDecay WITHOUT learning-rate scaling (BaseNetConfigDeserializer.java:231-236)
    regs = json.loads(conf.toJson())["confs"][0]["layer"]["regularization"]
    wd = [r for r in regs if r["@class"].endswith("WeightDecay")]
    assert len(wd) == 1 and wd[0]["applyLR"] is False and wd[0]["coeff"]["value"] == 3e-4
    # the upgraded document is current format: a second pass changes nothing, and it is no longer flagged
    cur = json.loads(conf.toJson())
    assert not LJ.is_legacy(cur) and LJ.upgrade(cur) == cur
    assert MultiLayerConfiguration.fromJson(conf.toJson()) == conf
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    net = MultiLayerNetwork(conf)
    net.init()
    y = net.output(torch.randn(5, 4))
    assert tuple(y.shape) == (5, 3) and torch.allclose(y.sum(1), torch.ones(5), atol=1e-5)


@pytest.mark.parametrize("name,cls,fields", [
    ("SGD", U.Sgd, {"learningRate": 0.2}),
    ("ADAM", U.Adam, {"learningRate": 0.01, "beta1": 0.8, "beta2": 0.95, "epsilon": 1e-7}),
    ("ADAMAX", U.AdaMax, {"learningRate": 0.01, "beta1": 0.8, "beta2": 0.95}),
    ("NADAM", U.Nadam, {"learningRate": 0.01, "beta1": 0.8, "beta2": 0.95}),
    ("ADADELTA", U.AdaDelta, {"rho": 0.9}),
    ("ADAGRAD", U.AdaGrad, {"learningRate": 0.3}),
    ("RMSPROP", U.RmsProp, {"learningRate": 0.02, "rmsDecay": 0.9}),
    ("NONE", U.NoOp, {})])
def test_legacy_updater_enum_maps_to_iupdater(name, cls, fields):
    on = {"updater": name, "learningRate": fields.get("learningRate", 0.1), "momentum": 0.9, "rho": fields.get("rho", 0.95),
          "rmsDecay": fields.get("rmsDecay", 0.95), "adamMeanDecay": fields.get("beta1", 0.9), "adamVarDecay": fields.get("beta2", 0.999)}
    if "epsilon" in fields:
        on["epsilon"] = fields["epsilon"]
    d = {"layer": {"dense": dict(on, nIn=2, nOut=2, activationFunction="tanh", weightInit="RELU")}}
    c = NeuralNetConfiguration.fromJson(json.dumps(d))
    u = c.layer.iupdater
    assert isinstance(u, cls)
    for k, v in fields.items():
        assert getattr(u, k) == v, k
    if name in ("ADAM", "ADAMAX", "NADAM") and "epsilon" not in fields:
        assert u.epsilon == 1e-8                                      # NaN / absent epsilon -> the updater's default
    assert c.layer.weightInitFn.scheme == "RELU" and type(c.layer.activationFn).__name__ == "ActivationTanH"


def test_100alpha_wrapper_objects_for_activations_losses_preprocessors_and_vertices():
    adam = {"@class": "org.nd4j.linalg.learning.config.Adam", "learningRate": 0.001, "beta1": 0.9, "beta2": 0.999, "epsilon": 1e-8}
    def nnc(layer):
        return {"layer": layer, "seed": 7, "miniBatch": True, "minimize": True, "optimizationAlgo": "STOCHASTIC_GRADIENT_DESCENT"}
    conv = {"convolution": {"nin": 1, "nout": 4, "kernelSize": [3, 3], "stride": [1, 1], "padding": [0, 0], "convolutionMode": "Truncate",
                            "activationFn": {"ReLU": {}}, "iupdater": adam, "weightInit": "XAVIER_UNIFORM", "layerName": "c1"}}
    pool = {"subsampling": {"poolingType": "MAX", "kernelSize": [2, 2], "stride": [2, 2], "padding": [0, 0], "convolutionMode": "Truncate",
                            "layerName": "p1"}}
    dense = {"dense": {"nin": 36, "nout": 8, "activationFn": {"LReLU": {"alpha": 0.2}}, "iupdater": adam, "weightInit": "DISTRIBUTION",
                       "dist": {"normal": {"mean": 0.0, "std": 0.05}}, "layerName": "d1"}}
    out = {"output": {"nin": 8, "nout": 2, "activationFn": {"Softmax": {}}, "lossFn": {"MCXENT": {"softmaxClipEps": 1e-10}}, "iupdater": adam,
                      "layerName": "out"}}
    old = {"vertices": {
               "c1": {"LayerVertex": {"layerConf": nnc(conv), "preProcessor": None}},
               "p1": {"LayerVertex": {"layerConf": nnc(pool), "preProcessor": None}},
               "d1": {"LayerVertex": {"layerConf": nnc(dense),
                                      "preProcessor": {"cnnToFeedForward": {"inputHeight": 3, "inputWidth": 3, "numChannels": 4}}}},
               "d2": {"LayerVertex": {"layerConf": nnc(json.loads(json.dumps(dense).replace('"nin": 36', '"nin": 8').replace("d1", "d2"))),
                                      "preProcessor": None}},
               "sum": {"ElementWiseVertex": {"op": "Add"}},
               "out": {"LayerVertex": {"layerConf": nnc(out), "preProcessor": None}}},
           "vertexInputs": {"c1": ["in"], "p1": ["c1"], "d1": ["p1"], "d2": ["d1"], "sum": ["d1", "d2"], "out": ["sum"]},
           "networkInputs": ["in"], "networkOutputs": ["out"], "backpropType": "Standard", "pretrain": False, "backprop": True,
           "defaultConfiguration": nnc(None)}
    assert LJ.is_legacy(old)
    conf = ComputationGraphConfiguration.fromJson(json.dumps(old))
    v = conf.vertices
    assert type(v["sum"]).__name__ == "ElementWiseVertex" and v["sum"].op == "Add"
    c1, d1, o = v["c1"].layerConf.layer, v["d1"].layerConf.layer, v["out"].layerConf.layer
    assert isinstance(c1, ConvolutionLayer) and type(c1.activationFn).__name__ == "ActivationReLU" and c1.weightInitFn.scheme == "XAVIER_UNIFORM"
    assert isinstance(v["p1"].layerConf.layer, SubsamplingLayer)
    assert type(d1.activationFn).__name__ == "ActivationLReLU" and abs(d1.activationFn.alpha - 0.2) < 1e-12
    assert d1.weightInitFn.scheme == "DISTRIBUTION" and abs(d1.weightInitFn.distribution.std - 0.05) < 1e-12
    assert type(v["d1"].preProcessor).__name__ == "CnnToFeedForwardPreProcessor" and v["d1"].preProcessor.numChannels == 4
    assert type(o.lossFn).__name__ == "LossMCXENT" and isinstance(o.iupdater, U.Adam)
    from deeplearning4j_amd.nn.graph.computation_graph import ComputationGraph
    net = ComputationGraph(conf)
    net.init()
    y = net.outputSingle(torch.randn(3, 1, 8, 8))
    assert tuple(y.shape) == (3, 2) and torch.isfinite(y).all()
    assert ComputationGraphConfiguration.fromJson(conf.toJson()) == conf


def test_legacy_recurrent_and_composable_preprocessor_wrappers():
    old = {"confs": [{"layer": {"LSTM": {"nin": 3, "nout": 5, "activationFn": {"TanH": {}}, "gateActivationFn": {"Sigmoid": {}},
                                          "forgetGateBiasInit": 1.0, "updater": "RMSPROP", "learningRate": 0.01, "rmsDecay": 0.9,
                                          "weightInit": "XAVIER"}}},
                     {"layer": {"rnnoutput": {"nin": 5, "nout": 2, "activationFn": {"Softmax": {}}, "lossFn": {"MCXENT": {}},
                                               "updater": "RMSPROP", "learningRate": 0.01, "rmsDecay": 0.9, "weightInit": "XAVIER"}}}],
           "inputPreProcessors": {"0": {"composableInput": {"inputPreProcessors": [{"feedForwardToRnn": {}}, {"rnnToFeedForward": {}},
                                                                                   {"feedForwardToRnn": {}}]}}},
           "backpropType": "TruncatedBPTT", "tbpttFwdLength": 7, "tbpttBackLength": 7}
    conf = MultiLayerConfiguration.fromJson(old)
    assert isinstance(conf.confs[0].layer, LSTM) and isinstance(conf.confs[1].layer, RnnOutputLayer)
    assert type(conf.confs[0].layer.gateActivationFn).__name__ == "ActivationSigmoid"
    assert isinstance(conf.confs[0].layer.iupdater, U.RmsProp) and conf.confs[0].layer.iupdater.rmsDecay == 0.9
    pre = conf.inputPreProcessors[0]
    assert type(pre).__name__ == "ComposableInputPreProcessor" and len(pre.inputPreProcessors) == 3
    assert conf.backpropType == "TruncatedBPTT" and conf.tbpttFwdLength == 7


def test_unknown_legacy_names_fail_loudly():
    with pytest.raises(ValueError):
        NeuralNetConfiguration.fromJson(json.dumps({"layer": {"dense": {"nIn": 1, "nOut": 1, "activationFunction": "nosuch"}}}))
    with pytest.raises(ValueError):
        NeuralNetConfiguration.fromJson(json.dumps({"layer": {"dense": {"nIn": 1, "nOut": 1, "updater": "FANCY", "learningRate": 0.1}}}))
    with pytest.raises(ValueError):
        NeuralNetConfiguration.fromJson(json.dumps({"layer": {"output": {"nIn": 1, "nOut": 1, "lossFunction": "NOPE", "activationFunction": "tanh"}}}))
