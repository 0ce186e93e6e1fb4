"""Parameter initializers (reference: nn/params/{DefaultParamInitializer, ConvolutionParamInitializer,
BatchNormalizationParamInitializer, LSTMParamInitializer, GravesLSTMParamInitializer, ...}.java) — the flat-vector layout of every
layer (order and shapes are the checkpoint ABI). The layouts themselves live with the layer configurations
(nn/conf/layers.py ``param_specs``); these classes expose them through the reference's ParamInitializer interface."""
from __future__ import annotations

from .conf import layers as C


class ParamInitializer:
    WEIGHT_KEY, BIAS_KEY = "W", "b"
    LAYER_TYPES = ()

    @classmethod
    def getInstance(cls):
        return cls()

    def _specs(self, conf):
        return conf.param_specs()

    def numParams(self, conf):
        return sum(s.numel for s in self._specs(conf))

    def paramKeys(self, conf):
        return [s.key for s in self._specs(conf)]

    def weightKeys(self, conf):
        return [s.key for s in self._specs(conf) if s.role in ("W", "RW")]

    def biasKeys(self, conf):
        return [s.key for s in self._specs(conf) if s.role == "b"]

    def isWeightParam(self, conf, key):
        return key in self.weightKeys(conf)

    def isBiasParam(self, conf, key):
        return key in self.biasKeys(conf)

    def paramShapes(self, conf):
        return {s.key: tuple(s.shape) for s in self._specs(conf)}

    def init(self, conf, paramsView, initializeParams=True):
        """Views of ``paramsView`` (flat, length numParams) per parameter key, in layout order ('f' order blocks are column-major)."""
        out, off = {}, 0
        flat = paramsView.reshape(-1)
        for s in self._specs(conf):
            v = flat[off:off + s.numel]
            out[s.key] = v.view(tuple(reversed(s.shape))).t() if (s.order == "f" and len(s.shape) == 2) else v.view(s.shape)
            off += s.numel
        return out

    def getGradientsFromFlattened(self, conf, gradientView):
        return self.init(conf, gradientView, False)


class DefaultParamInitializer(ParamInitializer):
    GAIN_KEY = "g"
    LAYER_TYPES = (C.DenseLayer, C.OutputLayer)


class ConvolutionParamInitializer(ParamInitializer):
    LAYER_TYPES = (C.ConvolutionLayer,)


class BatchNormalizationParamInitializer(ParamInitializer):
    GAMMA, BETA, GLOBAL_MEAN, GLOBAL_VAR, GLOBAL_LOG_STD = "gamma", "beta", "mean", "var", "log10stdev"
    LAYER_TYPES = (C.BatchNormalization,)


class LSTMParamInitializer(ParamInitializer):
    INPUT_WEIGHT_KEY, RECURRENT_WEIGHT_KEY, BIAS_KEY = "W", "RW", "b"
    LAYER_TYPES = (C.LSTM,)


class GravesLSTMParamInitializer(LSTMParamInitializer):
    LAYER_TYPES = (C.GravesLSTM,)


def initializer_for(conf) -> ParamInitializer:
    """The ParamInitializer a layer configuration would return from ``initializer()`` in the reference."""
    for cls in (GravesLSTMParamInitializer, LSTMParamInitializer, BatchNormalizationParamInitializer, ConvolutionParamInitializer,
                DefaultParamInitializer):
        if isinstance(conf, cls.LAYER_TYPES):
            return cls()
    return ParamInitializer()
