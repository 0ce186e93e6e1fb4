"""Legacy network-configuration JSON (releases up to 1.0.0-alpha and the 0.x line) -> current format.

Reference (behaviour, not code):
* deeplearning4j-nn/.../nn/conf/serde/legacy/LegacyJsonFormat.java:40-187 — up to 1.0.0-alpha polymorphic types were written as a
  WRAPPER OBJECT ``{"dense": {...}}`` with short names; afterwards as ``{"@class": "org...DenseLayer", ...}``. The six name tables
  below (layers, graph vertices, input pre-processors, activations, loss functions, reconstruction distributions) are the subtype
  names of its mix-ins.
* deeplearning4j-nn/.../nn/conf/serde/BaseNetConfigDeserializer.java:66-321 (and its MultiLayer / ComputationGraph subclasses) — field
  level upgrades applied after parsing: ``updater`` enum + learningRate / momentum / rho / rmsDecay / adamMeanDecay / adamVarDecay /
  epsilon -> IUpdater; ``l1`` / ``l2`` / ``l1Bias`` / ``l2Bias`` -> regularization lists (L2 as WeightDecay without LR scaling, the
  1.0.0-beta3 behaviour); ``weightInit`` enum (+ ``dist``) -> IWeightInit; ``activationFunction`` string -> activation object;
  ``lossFunction`` enum of 0.5.0 -> loss object; ``dropOut`` number -> Dropout (retain probability; 0 = none).

This module rewrites the parsed JSON tree into the current layout BEFORE the regular readers (configuration.py / layers.py) see it,
so there is one reader for every era. ``upgrade`` is idempotent: current-format input passes through unchanged.
"""
from __future__ import annotations

import copy
import json
from typing import Any, Dict, Optional

_L = "org.deeplearning4j.nn.conf.layers."
_G = "org.deeplearning4j.nn.conf.graph."
_P = "org.deeplearning4j.nn.conf.preprocessor."
_A = "org.nd4j.linalg.activations.impl."
_LF = "org.nd4j.linalg.lossfunctions.impl."
_V = "org.deeplearning4j.nn.conf.layers.variational."
_U = "org.nd4j.linalg.learning.config."
_R = "org.nd4j.linalg.learning.regularization."
_W = "org.deeplearning4j.nn.weights."

# wrapper-object names (LegacyJsonFormat.java mix-ins)
LAYER_NAMES = {
    "autoEncoder": _L + "AutoEncoder", "convolution": _L + "ConvolutionLayer", "convolution1d": _L + "Convolution1DLayer",
    "gravesLSTM": _L + "GravesLSTM", "LSTM": _L + "LSTM", "gravesBidirectionalLSTM": _L + "GravesBidirectionalLSTM",
    "output": _L + "OutputLayer", "CenterLossOutputLayer": _L + "CenterLossOutputLayer", "rnnoutput": _L + "RnnOutputLayer",
    "loss": _L + "LossLayer", "dense": _L + "DenseLayer", "subsampling": _L + "SubsamplingLayer",
    "subsampling1d": _L + "Subsampling1DLayer", "batchNormalization": _L + "BatchNormalization",
    "localResponseNormalization": _L + "LocalResponseNormalization", "embedding": _L + "EmbeddingLayer",
    "activation": _L + "ActivationLayer", "VariationalAutoencoder": _V + "VariationalAutoencoder", "dropout": _L + "DropoutLayer",
    "GlobalPooling": _L + "GlobalPoolingLayer", "zeroPadding": _L + "ZeroPaddingLayer", "zeroPadding1d": _L + "ZeroPadding1DLayer",
    "FrozenLayer": _L + "misc.FrozenLayer", "Upsampling2D": _L + "Upsampling2D", "Yolo2OutputLayer": _L + "objdetect.Yolo2OutputLayer",
    "RnnLossLayer": _L + "RnnLossLayer", "CnnLossLayer": _L + "CnnLossLayer", "Bidirectional": _L + "recurrent.Bidirectional",
    "SimpleRnn": _L + "recurrent.SimpleRnn", "ElementWiseMult": _L + "misc.ElementWiseMultiplicationLayer",
    "MaskLayer": _L + "util.MaskLayer", "MaskZeroLayer": _L + "util.MaskZeroLayer", "Cropping1D": _L + "convolutional.Cropping1D",
    "Cropping2D": _L + "convolutional.Cropping2D"}
VERTEX_NAMES = {n: _G + n for n in ("ElementWiseVertex", "MergeVertex", "SubsetVertex", "LayerVertex", "PreprocessorVertex", "StackVertex",
                                    "UnstackVertex", "L2Vertex", "ScaleVertex", "L2NormalizeVertex")}
VERTEX_NAMES.update({n: _G + "rnn." + n for n in ("LastTimeStepVertex", "ReverseTimeSeriesVertex", "DuplicateToTimeSeriesVertex")})
PREPROCESSOR_NAMES = {"cnnToFeedForward": _P + "CnnToFeedForwardPreProcessor", "cnnToRnn": _P + "CnnToRnnPreProcessor",
                      "composableInput": _P + "ComposableInputPreProcessor", "feedForwardToCnn": _P + "FeedForwardToCnnPreProcessor",
                      "feedForwardToRnn": _P + "FeedForwardToRnnPreProcessor", "rnnToFeedForward": _P + "RnnToFeedForwardPreProcessor",
                      "rnnToCnn": _P + "RnnToCnnPreProcessor"}
ACTIVATION_NAMES = {"Cube": "ActivationCube", "ELU": "ActivationELU", "HardSigmoid": "ActivationHardSigmoid", "HardTanh": "ActivationHardTanH",
                    "Identity": "ActivationIdentity", "LReLU": "ActivationLReLU", "RationalTanh": "ActivationRationalTanh",
                    "RectifiedTanh": "ActivationRectifiedTanh", "SELU": "ActivationSELU", "SWISH": "ActivationSwish", "ReLU": "ActivationReLU",
                    "RReLU": "ActivationRReLU", "Sigmoid": "ActivationSigmoid", "Softmax": "ActivationSoftmax", "SoftPlus": "ActivationSoftPlus",
                    "SoftSign": "ActivationSoftSign", "TanH": "ActivationTanH"}
ACTIVATION_NAMES = {k: _A + v for k, v in ACTIVATION_NAMES.items()}
LOSS_NAMES = {n: _LF + "Loss" + n for n in ("BinaryXENT", "CosineProximity", "Hinge", "KLD", "MAE", "L1", "MAPE", "MCXENT", "MSE", "L2", "MSLE",
                                            "NegativeLogLikelihood", "Poisson", "SquaredHinge", "FMeasure")}
RECON_NAMES = {"Gaussian": _V + "GaussianReconstructionDistribution", "Bernoulli": _V + "BernoulliReconstructionDistribution",
               "Exponential": _V + "ExponentialReconstructionDistribution", "Composite": _V + "CompositeReconstructionDistribution",
               "LossWrapper": _V + "LossFunctionWrapper"}

# 0.5.0 loss enum (BaseNetConfigDeserializer.handleLossBackwardCompatibility)
_LOSS_ENUM = {"MCXENT": "MCXENT", "MSE": "MSE", "NEGATIVELOGLIKELIHOOD": "NegativeLogLikelihood", "SQUARED_LOSS": "L2", "XENT": "BinaryXENT",
              "L1": "L1", "L2": "L2", "MEAN_ABSOLUTE_ERROR": "MAE", "MEAN_ABSOLUTE_PERCENTAGE_ERROR": "MAPE",
              "MEAN_SQUARED_LOGARITHMIC_ERROR": "MSLE", "KL_DIVERGENCE": "KLD", "COSINE_PROXIMITY": "CosineProximity", "HINGE": "Hinge",
              "SQUARED_HINGE": "SquaredHinge", "POISSON": "Poisson", "RECONSTRUCTION_CROSSENTROPY": "BinaryXENT"}
# "activationFunction": "softmax" of <= 0.7.1 (lower-cased lookup as getMap() in the reference)
_ACT_STRINGS = {"cube": "Cube", "elu": "ELU", "hardsigmoid": "HardSigmoid", "hardtanh": "HardTanh", "identity": "Identity", "linear": "Identity",
                "leakyrelu": "LReLU", "lrelu": "LReLU", "rationaltanh": "RationalTanh", "rectifiedtanh": "RectifiedTanh", "selu": "SELU",
                "swish": "SWISH", "relu": "ReLU", "rrelu": "RReLU", "sigmoid": "Sigmoid", "softmax": "Softmax", "softplus": "SoftPlus",
                "softsign": "SoftSign", "tanh": "TanH"}
_WEIGHT_INIT = {"ZERO": "WeightInitConstant", "ONES": "WeightInitConstant", "SIGMOID_UNIFORM": "WeightInitSigmoidUniform",
                "NORMAL": "WeightInitNormal", "LECUN_NORMAL": "WeightInitLecunNormal", "UNIFORM": "WeightInitUniform", "XAVIER": "WeightInitXavier",
                "XAVIER_UNIFORM": "WeightInitXavierUniform", "XAVIER_FAN_IN": "WeightInitXavierFanIn", "XAVIER_LEGACY": "WeightInitXavierLegacy",
                "RELU": "WeightInitRelu", "RELU_UNIFORM": "WeightInitReluUniform", "IDENTITY": "WeightInitIdentity",
                "LECUN_UNIFORM": "WeightInitLecunUniform", "VAR_SCALING_NORMAL_FAN_IN": "WeightInitVarScalingNormalFanIn",
                "VAR_SCALING_NORMAL_FAN_OUT": "WeightInitVarScalingNormalFanOut", "VAR_SCALING_NORMAL_FAN_AVG": "WeightInitVarScalingNormalFanAvg",
                "VAR_SCALING_UNIFORM_FAN_IN": "WeightInitVarScalingUniformFanIn", "VAR_SCALING_UNIFORM_FAN_OUT": "WeightInitVarScalingUniformFanOut",
                "VAR_SCALING_UNIFORM_FAN_AVG": "WeightInitVarScalingUniformFanAvg", "DISTRIBUTION": "WeightInitDistribution"}
_LEGACY_UPDATER_FIELDS = ("updater", "learningRate", "biasLearningRate", "momentum", "momentumSchedule", "rho", "rmsDecay", "adamMeanDecay",
                          "adamVarDecay", "epsilon", "learningRateSchedule", "lrPolicyDecayRate", "lrPolicySteps", "lrPolicyPower",
                          "learningRateDecayPolicy")
_DISTRIBUTIONS = {"normal": "NormalDistribution", "gaussian": "GaussianDistribution", "uniform": "UniformDistribution",
                  "binomial": "BinomialDistribution"}


def _unwrap(node: Any, table: Dict[str, str]) -> Any:
    """``{"dense": {...}}`` -> ``{"@class": ..., ...}`` when the single key is a name of ``table``; anything else is returned as is."""
    if isinstance(node, dict) and "@class" not in node and len(node) == 1:
        (k, v), = node.items()
        if k in table and (isinstance(v, dict) or v is None):
            out = {"@class": table[k]}
            out.update(v or {})
            return out
    return node


def is_legacy(d: Any) -> bool:
    """True when the tree contains a wrapper-object subtype or one of the legacy scalar fields."""
    found = [False]

    def walk(n, key=None):
        if found[0]:
            return
        if isinstance(n, dict):
            if "@class" not in n and len(n) == 1:
                k = next(iter(n))
                if key in ("layer", "fwd", "bwd", "underlying") and k in LAYER_NAMES or key == "activationFn" and k in ACTIVATION_NAMES \
                        or key in ("lossFn", "lossFunction") and k in LOSS_NAMES or k in VERTEX_NAMES and isinstance(n[k], dict) and key != "vertexInputs":
                    found[0] = True
                    return
            if "activationFunction" in n or ("updater" in n and isinstance(n["updater"], str)) or isinstance(n.get("weightInit"), str) \
                    or isinstance(n.get("lossFunction"), str) or isinstance(n.get("dropOut"), (int, float)):
                found[0] = True
                return
            for k, v in n.items():
                walk(v, k)
        elif isinstance(n, list):
            for v in n:
                walk(v, key)

    walk(d)
    return found[0]


def _fixed(v):
    return {"@class": "org.nd4j.linalg.schedule.FixedSchedule", "value": float(v)}


def _legacy_updater(on: Dict[str, Any]) -> Optional[Dict[str, Any]]:
    name = on.get("updater")
    if not isinstance(name, str):
        return None
    name = name.upper()
    lr = on.get("learningRate")
    eps = on.get("epsilon")
    nan = eps is None or (isinstance(eps, float) and eps != eps)

    def g(k, dflt):
        v = on.get(k)
        return dflt if v is None or (isinstance(v, float) and v != v) else v

    if name == "SGD":
        return {"@class": _U + "Sgd", "learningRate": g("learningRate", 1e-3)}
    if name in ("ADAM", "ADAMAX", "NADAM"):
        cls = {"ADAM": "Adam", "ADAMAX": "AdaMax", "NADAM": "Nadam"}[name]
        return {"@class": _U + cls, "learningRate": g("learningRate", 1e-3), "beta1": g("adamMeanDecay", 0.9), "beta2": g("adamVarDecay", 0.999),
                "epsilon": 1e-8 if nan else eps}
    if name == "ADADELTA":
        return {"@class": _U + "AdaDelta", "rho": g("rho", 0.95), "epsilon": 1e-6 if nan else eps}
    if name == "NESTEROVS":
        return {"@class": _U + "Nesterovs", "learningRate": g("learningRate", 0.1), "momentum": g("momentum", 0.9)}
    if name == "ADAGRAD":
        return {"@class": _U + "AdaGrad", "learningRate": g("learningRate", 0.1), "epsilon": 1e-6 if nan else eps}
    if name == "RMSPROP":
        return {"@class": _U + "RmsProp", "learningRate": g("learningRate", 0.1), "rmsDecay": g("rmsDecay", 0.95), "epsilon": 1e-8 if nan else eps}
    if name in ("NONE", "CUSTOM"):
        return {"@class": _U + "NoOp"}
    del lr
    raise ValueError("unknown legacy updater '%s'" % name)


def _legacy_distribution(d: Any) -> Any:
    """``{"normal": {"mean": 0, "std": 1}}`` (wrapper form) or an @class form -> @class form."""
    if isinstance(d, dict) and "@class" not in d and len(d) == 1:
        (k, v), = d.items()
        if k.lower() in _DISTRIBUTIONS:
            out = {"@class": "org.deeplearning4j.nn.conf.distribution." + _DISTRIBUTIONS[k.lower()]}
            out.update(v or {})
            return out
    return d


def _upgrade_layer(layer: Any) -> Any:
    layer = _unwrap(layer, LAYER_NAMES)
    if not isinstance(layer, dict):
        return layer
    on = layer
    # wrapped / nested layers
    for k in ("layer", "fwd", "bwd", "underlying"):
        if isinstance(on.get(k), dict):
            on[k] = _upgrade_layer(on[k])
    # activation: wrapper object, or the <= 0.7.1 string
    if "activationFn" in on:
        on["activationFn"] = _unwrap(on["activationFn"], ACTIVATION_NAMES)
    if on.get("activationFn") is None and isinstance(on.get("activationFunction"), str):
        key = _ACT_STRINGS.get(on["activationFunction"].lower())
        if key is None:
            raise ValueError("unknown legacy activation '%s'" % on["activationFunction"])
        on["activationFn"] = {"@class": ACTIVATION_NAMES[key]}
    on.pop("activationFunction", None)
    if "gateActivationFn" in on:
        on["gateActivationFn"] = _unwrap(on["gateActivationFn"], ACTIVATION_NAMES)
    # loss: wrapper object, or the 0.5.0 enum under "lossFunction"
    if "lossFn" in on:
        on["lossFn"] = _unwrap(on["lossFn"], LOSS_NAMES)
    lf = on.get("lossFunction")
    if isinstance(lf, dict):
        if on.get("lossFn") is None:
            on["lossFn"] = _unwrap(lf, LOSS_NAMES)
        on.pop("lossFunction")
    elif isinstance(lf, str):
        if on.get("lossFn") is None:
            if lf.upper() not in _LOSS_ENUM:
                raise ValueError("unknown legacy loss function '%s'" % lf)
            on["lossFn"] = {"@class": LOSS_NAMES[_LOSS_ENUM[lf.upper()]]}
        on.pop("lossFunction")
    if "reconstructionDistribution" in on:
        rd = _unwrap(on["reconstructionDistribution"], RECON_NAMES)
        if isinstance(rd, dict):
            if "activationFn" in rd:
                rd["activationFn"] = _unwrap(rd["activationFn"], ACTIVATION_NAMES)
            if "lossFunction" in rd and isinstance(rd["lossFunction"], dict):
                rd["lossFunction"] = _unwrap(rd["lossFunction"], LOSS_NAMES)
        on["reconstructionDistribution"] = rd
    if "pzxActivationFn" in on:
        on["pzxActivationFn"] = _unwrap(on["pzxActivationFn"], ACTIVATION_NAMES)
    # updater enum + hyper-parameters -> IUpdater
    if on.get("iupdater") is None and isinstance(on.get("updater"), str):
        on["iupdater"] = _legacy_updater(on)
        blr = on.get("biasLearningRate")
        if isinstance(blr, (int, float)) and blr == blr and on.get("learningRate") is not None and blr != on["learningRate"] \
                and on.get("biasUpdater") is None and "learningRate" in (on["iupdater"] or {}):
            bu = dict(on["iupdater"])
            bu["learningRate"] = blr
            on["biasUpdater"] = bu
    for k in _LEGACY_UPDATER_FIELDS:
        if k in on and not (k == "updater" and isinstance(on[k], dict)):
            on.pop(k)
    # l1 / l2 scalars -> regularization lists
    if any(k in on for k in ("l1", "l2", "l1Bias", "l2Bias")) and "regularization" not in on:
        reg, regb = [], []
        for key, dst in (("l1", reg), ("l1Bias", regb)):
            v = on.get(key)
            if isinstance(v, (int, float)) and v == v and v > 0.0:
                dst.append({"@class": _R + "L1Regularization", "l1": _fixed(v)})
        for key, dst in (("l2", reg), ("l2Bias", regb)):
            v = on.get(key)
            if isinstance(v, (int, float)) and v == v and v > 0.0:
                dst.append({"@class": _R + "WeightDecay", "coeff": _fixed(v), "applyLR": False})
        on["regularization"], on["regularizationBias"] = reg, regb
    for k in ("l1", "l2", "l1Bias", "l2Bias"):
        on.pop(k, None)
    # weight init enum (+ dist) -> IWeightInit
    wi = on.get("weightInit")
    if on.get("weightInitFn") is None and isinstance(wi, str):
        cls = _WEIGHT_INIT.get(wi.upper())
        if cls is None:
            raise ValueError("unknown legacy weight init '%s'" % wi)
        fn: Dict[str, Any] = {"@class": _W + cls}
        if wi.upper() == "DISTRIBUTION":
            fn["distribution"] = _legacy_distribution(on.get("dist"))
        elif wi.upper() in ("ZERO", "ONES"):
            fn["value"] = 0.0 if wi.upper() == "ZERO" else 1.0
        on["weightInitFn"] = fn
    on.pop("weightInit", None)
    on.pop("dist", None)
    # dropOut number -> Dropout (retain probability; 0 disables)
    if "dropOut" in on:
        p = on.pop("dropOut")
        if on.get("idropout") is None and isinstance(p, (int, float)) and p == p and 0.0 < p < 1.0:
            on["idropout"] = {"@class": "org.deeplearning4j.nn.conf.dropout.Dropout", "p": float(p), "pSchedule": None}
    # renamed scalar fields
    if "nIn" in on and "nin" not in on:
        on["nin"] = on.pop("nIn")
    if "nOut" in on and "nout" not in on:
        on["nout"] = on.pop("nOut")
    return on


def _upgrade_nnc(conf: Any) -> Any:
    """One NeuralNetConfiguration node: its layer; pre-0.8 configurations also carried network-wide learning-rate fields here."""
    if isinstance(conf, dict) and "layer" in conf:
        layer = conf["layer"]
        # 0.x: learning-rate / regularisation hyper-parameters on the CONFIGURATION level apply to a layer that has none of its own
        if isinstance(layer, dict):
            inner = layer if "@class" in layer else (next(iter(layer.values())) if len(layer) == 1 and isinstance(next(iter(layer.values())), dict) else None)
            if inner is not None:
                for k in ("learningRate", "momentum", "rho", "rmsDecay", "adamMeanDecay", "adamVarDecay", "epsilon", "l1", "l2", "updater"):
                    if k in conf and k not in inner and not isinstance(conf[k], dict):
                        inner[k] = conf[k]
        conf["layer"] = _upgrade_layer(layer)
        for k in ("learningRate", "momentum", "rho", "rmsDecay", "adamMeanDecay", "adamVarDecay", "epsilon", "l1", "l2", "updater",
                  "useRegularization", "useDropConnect", "numIterations", "learningRatePolicy", "lrPolicyDecayRate", "lrPolicySteps",
                  "lrPolicyPower", "leakyreluAlpha", "pretrain"):
            if not isinstance(conf.get(k), dict):
                conf.pop(k, None)
    return conf


def _upgrade_preprocessor(p: Any) -> Any:
    p = _unwrap(p, PREPROCESSOR_NAMES)
    if isinstance(p, dict) and isinstance(p.get("inputPreProcessors"), list):
        p["inputPreProcessors"] = [_upgrade_preprocessor(x) for x in p["inputPreProcessors"]]
    return p


def _upgrade_vertex(v: Any) -> Any:
    v = _unwrap(v, VERTEX_NAMES)
    if isinstance(v, dict):
        if isinstance(v.get("layerConf"), dict):
            v["layerConf"] = _upgrade_nnc(v["layerConf"])
        if v.get("preProcessor") is not None:
            v["preProcessor"] = _upgrade_preprocessor(v["preProcessor"])
    return v


def upgrade(d: Any) -> Any:
    """Rewrite a parsed MultiLayerConfiguration / ComputationGraphConfiguration / NeuralNetConfiguration JSON tree of any era
    into the current format (a deep copy; the argument is not modified)."""
    if isinstance(d, (str, bytes)):
        d = json.loads(d)
    d = copy.deepcopy(d)
    if not isinstance(d, dict):
        return d
    if "confs" in d:
        d["confs"] = [_upgrade_nnc(c) for c in d["confs"]]
        if isinstance(d.get("inputPreProcessors"), dict):
            d["inputPreProcessors"] = {k: _upgrade_preprocessor(v) for k, v in d["inputPreProcessors"].items()}
        if "pretrain" in d or "backprop" in d:                   # removed switches of the 0.x line
            d.pop("pretrain", None)
            d.pop("backprop", None)
    elif "vertices" in d:
        d["vertices"] = type(d["vertices"])((k, _upgrade_vertex(v)) for k, v in d["vertices"].items())
        if isinstance(d.get("defaultConfiguration"), dict):
            d["defaultConfiguration"] = _upgrade_nnc(d["defaultConfiguration"])
        d.pop("pretrain", None)
        d.pop("backprop", None)
    elif "layer" in d:
        d = _upgrade_nnc(d)
    return d
