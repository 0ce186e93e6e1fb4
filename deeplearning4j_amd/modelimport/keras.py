"""Keras model import (reference: deeplearning4j-modelimport/.../keras/{KerasModelImport,KerasModel,KerasSequentialModel}.java,
keras/layers/**/Keras*.java (~60 layer mappers), keras/utils/{KerasLayerUtils,KerasActivationUtils,KerasInitilizationUtils,
KerasLossUtils}.java, Hdf5Archive.java).

Supported inputs: the model JSON (``model.to_json()`` / the ``model_config`` attribute) for Sequential and Functional models, with
weights from an ``.npz`` / dict keyed ``<layer>/<weight>`` (e.g. ``dense_1/kernel``), or from an ``.h5`` file through
``modelimport.hdf5.Hdf5Archive`` when that module can parse it.  Networks are NCHW: ``channels_last`` Keras kernels are
converted (HWIO -> OIHW) and a Dense that follows a Flatten gets its input rows permuted from (h, w, c) to (c, h, w) order, so
the imported network consumes NCHW input and produces the same outputs as the Keras model on the NHWC-transposed input.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..nn.conf.base import ConvolutionMode, InputType, PoolingType
from ..nn.conf.configuration import NeuralNetConfiguration
from ..nn.conf.graph import ElementWiseVertex, MergeVertex
from ..nn.conf import layers as L


class InvalidKerasConfigurationException(ValueError):
    pass


class UnsupportedKerasConfigurationException(NotImplementedError):
    pass


_ACT = {"linear": "IDENTITY", "relu": "RELU", "relu6": "RELU6", "sigmoid": "SIGMOID", "tanh": "TANH", "softmax": "SOFTMAX", "softplus": "SOFTPLUS",
        "softsign": "SOFTSIGN", "elu": "ELU", "selu": "SELU", "hard_sigmoid": "HARDSIGMOID", "swish": "SWISH", "silu": "SWISH", "gelu": "GELU",
        "leaky_relu": "LEAKYRELU", "mish": "MISH", "exponential": None}
_INIT = {"glorot_uniform": "XAVIER_UNIFORM", "glorot_normal": "XAVIER", "he_normal": "RELU", "he_uniform": "RELU_UNIFORM", "lecun_normal": "LECUN_NORMAL",
         "lecun_uniform": "LECUN_UNIFORM", "zeros": "ZERO", "ones": "ONES", "identity": "IDENTITY", "random_uniform": "UNIFORM", "random_normal": "NORMAL",
         "truncated_normal": "NORMAL", "orthogonal": "XAVIER", "variance_scaling": "VAR_SCALING_NORMAL_FAN_IN",
         "GlorotUniform": "XAVIER_UNIFORM", "GlorotNormal": "XAVIER", "HeNormal": "RELU", "HeUniform": "RELU_UNIFORM", "Zeros": "ZERO", "Ones": "ONES",
         "Orthogonal": "XAVIER", "RandomNormal": "NORMAL", "RandomUniform": "UNIFORM", "TruncatedNormal": "NORMAL", "VarianceScaling": "VAR_SCALING_NORMAL_FAN_IN"}
_LOSS = {"mean_squared_error": "MSE", "mse": "MSE", "mean_absolute_error": "MEAN_ABSOLUTE_ERROR", "mae": "MEAN_ABSOLUTE_ERROR",
         "mean_absolute_percentage_error": "MEAN_ABSOLUTE_PERCENTAGE_ERROR", "mape": "MEAN_ABSOLUTE_PERCENTAGE_ERROR",
         "mean_squared_logarithmic_error": "MEAN_SQUARED_LOGARITHMIC_ERROR", "msle": "MEAN_SQUARED_LOGARITHMIC_ERROR", "squared_hinge": "SQUARED_HINGE",
         "hinge": "HINGE", "binary_crossentropy": "XENT", "categorical_crossentropy": "MCXENT", "sparse_categorical_crossentropy": "SPARSE_MCXENT",
         "kullback_leibler_divergence": "KL_DIVERGENCE", "kld": "KL_DIVERGENCE", "poisson": "POISSON", "cosine_proximity": "COSINE_PROXIMITY",
         "cosine_similarity": "COSINE_PROXIMITY"}


def map_activation(name):
    if isinstance(name, dict):
        name = name.get("class_name") or name.get("config", {}).get("name")
    if name is None:
        return "IDENTITY"
    a = _ACT.get(str(name).lower(), "?")
    if a in ("?", None):
        raise UnsupportedKerasConfigurationException("Unknown Keras activation function %s" % name)
    return a


def map_init(spec):
    if spec is None:
        return "XAVIER"
    name = spec.get("class_name") if isinstance(spec, dict) else spec
    return _INIT.get(name, _INIT.get(str(name).lower(), "XAVIER"))


def map_loss(name):
    if isinstance(name, dict):
        name = name.get("class_name") or name.get("config", {}).get("name")
    key = "".join(("_" + ch.lower() if ch.isupper() else ch) for ch in str(name)).lstrip("_") if any(ch.isupper() for ch in str(name)) else str(name)
    if key not in _LOSS:
        raise UnsupportedKerasConfigurationException("Unknown Keras loss function " + str(name))
    return _LOSS[key]


def _pair(v, n=2):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


def _reg(cfg, key):
    r = cfg.get(key)
    if not r:
        return None, None
    c = r.get("config", r)
    return (c.get("l1") or None), (c.get("l2") or None)


class _Mapped:
    """Result of mapping one Keras layer: a DL4J layer (or vertex / None for structural layers) + weight conversion."""

    def __init__(self, layer=None, vertex=None, weights=None, passthrough=False, flatten=False, input_shape=None):
        self.layer, self.vertex, self.weights, self.passthrough, self.flatten, self.input_shape = layer, vertex, weights, passthrough, flatten, input_shape


class KerasLayerMapper:
    def __init__(self, enforceTrainingConfig=False):
        self.enforce = enforceTrainingConfig

    # shapes flow as Keras output shapes without the batch dim, e.g. (h, w, c) or (t, f) or (f,)
    def map(self, cls, cfg, in_shape, channels_last=True) -> Tuple[_Mapped, tuple]:
        fn = getattr(self, "_" + cls, None)
        if fn is None:
            raise UnsupportedKerasConfigurationException("Unsupported keras layer type " + cls)
        return fn(cfg, in_shape)

    @staticmethod
    def _common(cfg):
        kw = {"name": cfg.get("name")}
        l1, l2 = _reg(cfg, "kernel_regularizer")
        if l1: kw["l1"] = l1
        if l2: kw["l2"] = l2
        l1b, l2b = _reg(cfg, "bias_regularizer")
        if l1b: kw["l1Bias"] = l1b
        if l2b: kw["l2Bias"] = l2b
        if "kernel_initializer" in cfg:
            kw["weightInit"] = map_init(cfg["kernel_initializer"])
        return kw

    # -- core
    def _InputLayer(self, cfg, s):
        shp = cfg.get("batch_input_shape") or cfg.get("batch_shape")
        return _Mapped(passthrough=True, input_shape=tuple(shp[1:]) if shp else None), tuple(shp[1:]) if shp else s

    def _Dense(self, cfg, s):
        kw = self._common(cfg)
        n = int(cfg["units"])
        layer = L.DenseLayer(nOut=n, activation=map_activation(cfg.get("activation")), hasBias=cfg.get("use_bias", True), **kw)
        if len(s) == 2:                      # Dense on [t, f] acts per time step
            layer = L.TimeDistributed(layer)
        return _Mapped(layer, weights=lambda w, ctx: {"W": w["kernel"], **({"b": w["bias"]} if "bias" in w else {})}), tuple(s[:-1]) + (n,)

    def _Activation(self, cfg, s):
        return _Mapped(L.ActivationLayer(activation=map_activation(cfg.get("activation")), name=cfg.get("name"))), s

    def _LeakyReLU(self, cfg, s):
        a = cfg.get("alpha", cfg.get("negative_slope", 0.3))
        return _Mapped(L.ActivationLayer(activation={"@class": "org.nd4j.linalg.activations.impl.ActivationLReLU", "alpha": float(a)}, name=cfg.get("name"))), s

    def _ELU(self, cfg, s):
        return _Mapped(L.ActivationLayer(activation="ELU", name=cfg.get("name"))), s

    def _ReLU(self, cfg, s):
        if cfg.get("max_value") in (6, 6.0):
            return _Mapped(L.ActivationLayer(activation="RELU6", name=cfg.get("name"))), s
        return _Mapped(L.ActivationLayer(activation="RELU", name=cfg.get("name"))), s

    def _Softmax(self, cfg, s):
        return _Mapped(L.ActivationLayer(activation="SOFTMAX", name=cfg.get("name"))), s

    def _PReLU(self, cfg, s):
        shp = tuple(s[::-1][:1] + s[:-1]) if len(s) == 3 else tuple(s)     # (h,w,c) -> (c,h,w)
        layer = L.PReLULayer(inputShape=shp, name=cfg.get("name"))

        def conv(w, ctx):
            a = w["alpha"]
            return {"W": np.transpose(a, (2, 0, 1)) if a.ndim == 3 else a}
        return _Mapped(layer, weights=conv), s

    def _Dropout(self, cfg, s):
        return _Mapped(L.DropoutLayer(dropOut=1.0 - float(cfg.get("rate", 0.5)), name=cfg.get("name"))), s      # DL4J keeps-probability

    _SpatialDropout1D = _SpatialDropout2D = _SpatialDropout3D = _Dropout
    _AlphaDropout = _GaussianDropout = _Dropout

    def _GaussianNoise(self, cfg, s):
        return _Mapped(L.DropoutLayer(dropOut=L.GaussianNoise(float(cfg.get("stddev", 0.1))), name=cfg.get("name"))), s

    def _Flatten(self, cfg, s):
        return _Mapped(passthrough=True, flatten=True), (int(np.prod(s)),)

    def _Reshape(self, cfg, s):
        tgt = tuple(int(v) for v in cfg["target_shape"])
        return _Mapped(passthrough=True, input_shape=None), tgt

    def _Masking(self, cfg, s):
        return _Mapped(L.MaskZeroLayer(L.ActivationLayer(activation="IDENTITY"), maskingValue=float(cfg.get("mask_value", 0.0)), name=cfg.get("name"))), s

    def _RepeatVector(self, cfg, s):
        n = int(cfg["n"])
        return _Mapped(L.RepeatVector(n=n, name=cfg.get("name"))), (n,) + tuple(s)

    def _Embedding(self, cfg, s):
        kw = {"name": cfg.get("name")}
        n_in, n_out = int(cfg["input_dim"]), int(cfg["output_dim"])
        t = s[0] if s else (cfg.get("input_length") or 1)
        layer = L.EmbeddingSequenceLayer(nIn=n_in, nOut=n_out, inputLength=int(t or 1), hasBias=False, **kw)
        return _Mapped(layer, weights=lambda w, ctx: {"W": w["embeddings"]}), (t, n_out)

    # -- convolution family
    @staticmethod
    def _mode(cfg):
        p = cfg.get("padding", "valid")
        if p == "same":
            return ConvolutionMode.Same
        if p == "valid":
            return ConvolutionMode.Truncate
        raise UnsupportedKerasConfigurationException("Unsupported keras padding mode " + str(p))

    @staticmethod
    def _out_len(n, k, s, mode, d=1):
        if n is None:
            return None
        if mode == ConvolutionMode.Same:
            return -(-n // s)
        return (n - d * (k - 1) - 1) // s + 1

    def _Conv2D(self, cfg, s, cls=L.ConvolutionLayer, depth_mult=None):
        if cfg.get("data_format", "channels_last") != "channels_last":
            raise UnsupportedKerasConfigurationException("channels_first Keras models are not supported")
        k, st, d = _pair(cfg["kernel_size"]), _pair(cfg.get("strides", 1)), _pair(cfg.get("dilation_rate", 1))
        mode = self._mode(cfg)
        kw = self._common(cfg)
        n = int(cfg.get("filters") or 0)
        extra = {}
        if depth_mult is not None:
            extra["depthMultiplier"] = int(depth_mult)
            if cls is L.DepthwiseConvolution2D:
                n = s[2] * int(depth_mult)
        layer = cls(kernelSize=k, stride=st, dilation=d, convolutionMode=mode, nOut=n, hasBias=cfg.get("use_bias", True),
                    activation=map_activation(cfg.get("activation")), **kw, **extra)

        def conv(w, ctx):
            out = {}
            if cls is L.SeparableConvolution2D:
                out["W"] = np.transpose(w["depthwise_kernel"], (3, 2, 0, 1))        # [kh,kw,in,mult] -> [mult,in,kh,kw]
                out["pW"] = np.transpose(w["pointwise_kernel"], (3, 2, 0, 1))       # [1,1,in*mult,out] -> [out,in*mult,1,1]
            elif cls is L.DepthwiseConvolution2D:
                out["W"] = w["depthwise_kernel"]                                    # DL4J depthwise layout is [kh,kw,in,mult] too
            elif cls is L.Deconvolution2D:
                out["W"] = np.transpose(w["kernel"], (3, 2, 0, 1))                  # [kh,kw,out,in] -> [in,out,kh,kw]
            else:
                out["W"] = np.transpose(w["kernel"], (3, 2, 0, 1))                  # HWIO -> OIHW
            if "bias" in w:
                out["b"] = w["bias"]
            return out
        if cls is L.Deconvolution2D:
            oh = None if s[0] is None else (s[0] * st[0] if mode == ConvolutionMode.Same else (s[0] - 1) * st[0] + k[0])
            ow = None if s[1] is None else (s[1] * st[1] if mode == ConvolutionMode.Same else (s[1] - 1) * st[1] + k[1])
        else:
            oh, ow = self._out_len(s[0], k[0], st[0], mode, d[0]), self._out_len(s[1], k[1], st[1], mode, d[1])
        return _Mapped(layer, weights=conv), (oh, ow, n)

    _Convolution2D = _Conv2D

    def _SeparableConv2D(self, cfg, s):
        return self._Conv2D(cfg, s, L.SeparableConvolution2D, cfg.get("depth_multiplier", 1))

    def _DepthwiseConv2D(self, cfg, s):
        return self._Conv2D(cfg, s, L.DepthwiseConvolution2D, cfg.get("depth_multiplier", 1))

    def _Conv2DTranspose(self, cfg, s):
        return self._Conv2D(cfg, s, L.Deconvolution2D)

    def _Conv1D(self, cfg, s):
        k, st, d = int(_pair(cfg["kernel_size"], 1)[0]), int(_pair(cfg.get("strides", 1), 1)[0]), int(_pair(cfg.get("dilation_rate", 1), 1)[0])
        if cfg.get("padding") == "causal":
            mode = ConvolutionMode.Causal if hasattr(ConvolutionMode, "Causal") else ConvolutionMode.Same
        else:
            mode = self._mode(cfg)
        n = int(cfg["filters"])
        layer = L.Convolution1DLayer(kernelSize=(k,), stride=(st,), dilation=(d,), convolutionMode=mode, nOut=n, hasBias=cfg.get("use_bias", True),
                                     activation=map_activation(cfg.get("activation")), **self._common(cfg))

        def conv(w, ctx):
            out = {"W": np.transpose(w["kernel"], (2, 1, 0))[:, :, :, None]}         # [k,in,out] -> [out,in,k,1]
            if "bias" in w:
                out["b"] = w["bias"]
            return out
        return _Mapped(layer, weights=conv), (self._out_len(s[0], k, st, mode, d), n)

    _Convolution1D = _Conv1D

    def _pool(self, cfg, s, kind, nd):
        k = _pair(cfg.get("pool_size", 2), nd)
        st = _pair(cfg.get("strides") or k, nd)
        mode = self._mode(cfg)
        if nd == 1:
            layer = L.Subsampling1DLayer(poolingType=kind, kernelSize=(int(k[0]),), stride=(int(st[0]),), convolutionMode=mode, name=cfg.get("name"))
            return _Mapped(layer), (self._out_len(s[0], k[0], st[0], mode), s[1])
        if nd == 3:
            layer = L.Subsampling3DLayer(poolingType=kind, kernelSize=k, stride=st, convolutionMode=mode, name=cfg.get("name"))
            return _Mapped(layer), tuple(self._out_len(s[i], k[i], st[i], mode) for i in range(3)) + (s[3],)
        layer = L.SubsamplingLayer(poolingType=kind, kernelSize=k, stride=st, convolutionMode=mode, name=cfg.get("name"))
        return _Mapped(layer), (self._out_len(s[0], k[0], st[0], mode), self._out_len(s[1], k[1], st[1], mode), s[2])

    def _MaxPooling2D(self, cfg, s): return self._pool(cfg, s, PoolingType.MAX, 2)
    def _AveragePooling2D(self, cfg, s): return self._pool(cfg, s, PoolingType.AVG, 2)
    def _MaxPooling1D(self, cfg, s): return self._pool(cfg, s, PoolingType.MAX, 1)
    def _AveragePooling1D(self, cfg, s): return self._pool(cfg, s, PoolingType.AVG, 1)
    def _MaxPooling3D(self, cfg, s): return self._pool(cfg, s, PoolingType.MAX, 3)
    def _AveragePooling3D(self, cfg, s): return self._pool(cfg, s, PoolingType.AVG, 3)

    def _gpool(self, cfg, s, kind):
        return _Mapped(L.GlobalPoolingLayer(poolingType=kind, name=cfg.get("name"))), (s[-1],)

    def _GlobalMaxPooling2D(self, cfg, s): return self._gpool(cfg, s, PoolingType.MAX)
    def _GlobalAveragePooling2D(self, cfg, s): return self._gpool(cfg, s, PoolingType.AVG)
    _GlobalMaxPooling1D, _GlobalAveragePooling1D = _GlobalMaxPooling2D, _GlobalAveragePooling2D
    _GlobalMaxPooling3D, _GlobalAveragePooling3D = _GlobalMaxPooling2D, _GlobalAveragePooling2D

    def _ZeroPadding2D(self, cfg, s):
        p = cfg.get("padding", 1)
        if isinstance(p, int):
            t = b = l = r = p
        elif isinstance(p[0], (list, tuple)):
            (t, b), (l, r) = p
        else:
            t = b = p[0]
            l = r = p[1]
        return _Mapped(L.ZeroPaddingLayer(padding=(t, b, l, r), name=cfg.get("name"))), (s[0] + t + b, s[1] + l + r, s[2])

    def _Cropping2D(self, cfg, s):
        p = cfg.get("cropping", 0)
        if isinstance(p, int):
            t = b = l = r = p
        elif isinstance(p[0], (list, tuple)):
            (t, b), (l, r) = p
        else:
            t = b = p[0]
            l = r = p[1]
        return _Mapped(L.Cropping2D(cropping=(t, b, l, r), name=cfg.get("name"))), (s[0] - t - b, s[1] - l - r, s[2])

    def _UpSampling2D(self, cfg, s):
        sz = _pair(cfg.get("size", 2))
        return _Mapped(L.Upsampling2D(size=sz, name=cfg.get("name"))), (s[0] * sz[0], s[1] * sz[1], s[2])

    def _UpSampling1D(self, cfg, s):
        sz = int(cfg.get("size", 2))
        return _Mapped(L.Upsampling1D(size=(sz,), name=cfg.get("name"))), (s[0] * sz, s[1])

    def _ZeroPadding1D(self, cfg, s):
        p = _pair(cfg.get("padding", 1))
        return _Mapped(L.ZeroPadding1DLayer(padding=p, name=cfg.get("name"))), (s[0] + p[0] + p[1], s[1])

    def _Cropping1D(self, cfg, s):
        p = _pair(cfg.get("cropping", 1))
        return _Mapped(L.Cropping1D(cropping=p, name=cfg.get("name"))), (s[0] - p[0] - p[1], s[1])

    # -- 3-D convolutional family (KerasConvolution3D / Deconvolution3D / ZeroPadding3D / Cropping3D / Upsampling3D)
    def _Conv3D(self, cfg, s, cls=None):
        if cfg.get("data_format", "channels_last") != "channels_last":
            raise UnsupportedKerasConfigurationException("channels_first Keras models are not supported")
        cls = cls or L.Convolution3D
        k, st, d = _pair(cfg["kernel_size"], 3), _pair(cfg.get("strides", 1), 3), _pair(cfg.get("dilation_rate", 1), 3)
        mode = self._mode(cfg)
        n = int(cfg["filters"])
        layer = cls(kernelSize=k, stride=st, dilation=d, convolutionMode=mode, nOut=n, hasBias=cfg.get("use_bias", True),
                    activation=map_activation(cfg.get("activation")), **self._common(cfg))

        def conv(w, ctx):
            # keras [kd, kh, kw, in, out] (transpose conv: [kd, kh, kw, out, in]) -> [out, in, kd, kh, kw] ([in, out, ...])
            out = {"W": np.transpose(w["kernel"], (4, 3, 0, 1, 2))}
            if "bias" in w:
                out["b"] = w["bias"]
            return out
        if cls is L.Deconvolution3D:
            dims = tuple(None if s[i] is None else (s[i] * st[i] if mode == ConvolutionMode.Same else (s[i] - 1) * st[i] + k[i]) for i in range(3))
        else:
            dims = tuple(self._out_len(s[i], k[i], st[i], mode, d[i]) for i in range(3))
        return _Mapped(layer, weights=conv), dims + (n,)

    _Convolution3D = _Conv3D

    def _Conv3DTranspose(self, cfg, s):
        return self._Conv3D(cfg, s, L.Deconvolution3D)

    _Deconvolution3D = _Conv3DTranspose

    @staticmethod
    def _six(p):
        if isinstance(p, int):
            return [p] * 6
        p = list(p)
        if p and isinstance(p[0], (list, tuple)):
            return [int(v) for pair in p for v in pair]
        return [int(v) for d in p for v in (d, d)]

    def _ZeroPadding3D(self, cfg, s):
        p = self._six(cfg.get("padding", 1))
        return _Mapped(L.ZeroPadding3DLayer(padding=tuple(p), name=cfg.get("name"))), (s[0] + p[0] + p[1], s[1] + p[2] + p[3], s[2] + p[4] + p[5], s[3])

    def _Cropping3D(self, cfg, s):
        p = self._six(cfg.get("cropping", 1))
        return _Mapped(L.Cropping3D(cropping=tuple(p), name=cfg.get("name"))), (s[0] - p[0] - p[1], s[1] - p[2] - p[3], s[2] - p[4] - p[5], s[3])

    def _UpSampling3D(self, cfg, s):
        sz = _pair(cfg.get("size", 2), 3)
        return _Mapped(L.Upsampling3D(size=tuple(sz), name=cfg.get("name"))), (s[0] * sz[0], s[1] * sz[1], s[2] * sz[2], s[3])

    # -- Keras 1 dilated convolutions (KerasAtrousConvolution1D / 2D): the same layers with atrous_rate as the dilation
    def _AtrousConvolution2D(self, cfg, s):
        c = dict(cfg)
        c.setdefault("filters", cfg.get("nb_filter"))
        c.setdefault("kernel_size", (cfg.get("nb_row", 3), cfg.get("nb_col", 3)))
        c.setdefault("strides", cfg.get("subsample", 1))
        c["dilation_rate"] = cfg.get("atrous_rate", cfg.get("dilation_rate", 1))
        c.setdefault("padding", cfg.get("border_mode", "valid"))
        return self._Conv2D(c, s)

    def _AtrousConvolution1D(self, cfg, s):
        c = dict(cfg)
        c.setdefault("filters", cfg.get("nb_filter"))
        c.setdefault("kernel_size", cfg.get("filter_length", 3))
        c.setdefault("strides", cfg.get("subsample_length", 1))
        c["dilation_rate"] = cfg.get("atrous_rate", cfg.get("dilation_rate", 1))
        c.setdefault("padding", cfg.get("border_mode", "valid"))
        return self._Conv1D(c, s)

    def _SpaceToDepth(self, cfg, s):
        b = int(cfg.get("block_size", 2))
        return _Mapped(L.SpaceToDepthLayer(blockSize=b, name=cfg.get("name"))), (s[0] // b, s[1] // b, s[2] * b * b)

    # -- locally connected (KerasLocallyConnected1D / 2D): unshared weights per output position
    def _LocallyConnected2D(self, cfg, s):
        k, st = _pair(cfg["kernel_size"]), _pair(cfg.get("strides", 1))
        if cfg.get("padding", "valid") != "valid":
            raise UnsupportedKerasConfigurationException("LocallyConnected2D supports only 'valid' padding")
        n = int(cfg["filters"])
        layer = L.LocallyConnected2D(kernelSize=k, stride=st, convolutionMode=ConvolutionMode.Truncate, nOut=n, hasBias=cfg.get("use_bias", True),
                                     activation=map_activation(cfg.get("activation")), **self._common(cfg))
        oh, ow = self._out_len(s[0], k[0], st[0], ConvolutionMode.Truncate), self._out_len(s[1], k[1], st[1], ConvolutionMode.Truncate)
        cin = s[2]

        def conv(w, ctx):
            # keras kernel [oh*ow, kh*kw*in, out] with the feature axis ordered (kh, kw, in); DL4J orders it (in, kh, kw)
            kk = w["kernel"].reshape(oh * ow, k[0], k[1], cin, n)
            out = {"W": np.transpose(kk, (0, 3, 1, 2, 4)).reshape(oh * ow, cin * k[0] * k[1], n)}
            if "bias" in w:
                b = w["bias"]
                out["b"] = b.reshape(-1, n).mean(0) if b.ndim > 1 and not ctx.get("per_position_bias") else b
            return out
        return _Mapped(layer, weights=conv), (oh, ow, n)

    def _LocallyConnected1D(self, cfg, s):
        k, st = int(_pair(cfg["kernel_size"], 1)[0]), int(_pair(cfg.get("strides", 1), 1)[0])
        if cfg.get("padding", "valid") != "valid":
            raise UnsupportedKerasConfigurationException("LocallyConnected1D supports only 'valid' padding")
        n = int(cfg["filters"])
        layer = L.LocallyConnected1D(kernelSize=(k,), stride=(st,), convolutionMode=ConvolutionMode.Truncate, nOut=n, hasBias=cfg.get("use_bias", True),
                                     activation=map_activation(cfg.get("activation")), **self._common(cfg))
        ot = self._out_len(s[0], k, st, ConvolutionMode.Truncate)
        cin = s[1]

        def conv(w, ctx):
            kk = w["kernel"].reshape(ot, k, cin, n)                          # (k, in) -> (in, k)
            out = {"W": np.transpose(kk, (0, 2, 1, 3)).reshape(ot, cin * k, n)}
            if "bias" in w:
                b = w["bias"]
                out["b"] = b.reshape(-1, n).mean(0) if b.ndim > 1 else b
            return out
        return _Mapped(layer, weights=conv), (ot, n)

    # -- custom layers shipped with the reference importer (layers/custom: KerasLRN, KerasPoolHelper)
    def _LRN(self, cfg, s):
        layer = L.LocalResponseNormalization(alpha=float(cfg.get("alpha", 1e-4)), beta=float(cfg.get("beta", 0.75)), k=float(cfg.get("k", 2)),
                                             n=float(cfg.get("n", 5)), name=cfg.get("name"))
        return _Mapped(layer), s

    _LRN2D = _LRN

    def _PoolHelper(self, cfg, s):
        """GoogLeNet helper: drops the first row and column of the feature map."""
        return _Mapped(L.Cropping2D(cropping=(1, 0, 1, 0), name=cfg.get("name"))), (s[0] - 1, s[1] - 1, s[2])

    def _ThresholdedReLU(self, cfg, s):
        th = float(cfg.get("theta", 1.0))
        return _Mapped(L.ActivationLayer(activation={"@class": "org.nd4j.linalg.activations.impl.ActivationThresholdedReLU", "theta": th},
                                         name=cfg.get("name"))), s

    def _Permute(self, cfg, s):
        """KerasPermute: a permutation of the non-batch axes, executed as a parameter-free lambda layer on the network's own layout
        (recurrent activations are [batch, features, time] inside the network, Keras sees [time, features])."""
        dims = tuple(int(d) for d in cfg["dims"])
        if len(s) == 2 and dims == (2, 1):
            fn = lambda x: x.transpose(1, 2)                                  # noqa: E731
            ot = lambda it: InputType.recurrent(it.timeSeriesLength, it.size)  # noqa: E731
            return _Mapped(L.SameDiffLambdaLayer(fn=fn, outputTypeFn=ot, name=cfg.get("name"))), (s[1], s[0])
        if dims == tuple(range(1, len(s) + 1)):
            return _Mapped(passthrough=True), s
        raise UnsupportedKerasConfigurationException("Permute%s is only supported for 2-D (time, feature) inputs" % (dims,))

    # -- Lambda layers: user-registered callables by layer name (KerasLayer.registerLambdaLayer)
    _LAMBDAS: Dict[str, object] = {}

    @staticmethod
    def registerLambdaLayer(name, fn_or_layer, outputShapeFn=None):
        KerasLayerMapper._LAMBDAS[name] = (fn_or_layer, outputShapeFn)

    @staticmethod
    def clearLambdaLayers():
        KerasLayerMapper._LAMBDAS.clear()

    def _Lambda(self, cfg, s):
        name = cfg.get("name")
        if name not in self._LAMBDAS:
            raise UnsupportedKerasConfigurationException(
                "No SameDiff Lambda layer found for Lambda layer %s. You can register a SameDiff Lambda layer using "
                "KerasLayerMapper.registerLambdaLayer(lambdaLayerName, sameDiffLambdaLayer)" % name)
        fn, shape_fn = self._LAMBDAS[name]
        layer = fn if isinstance(fn, L.Layer) else L.SameDiffLambdaLayer(fn=fn, name=name)
        return _Mapped(layer), (tuple(shape_fn(s)) if shape_fn else s)

    # -- Keras 1 Merge layer (KerasMerge): mode -> vertex
    def _Merge(self, cfg, s):
        mode = str(cfg.get("mode", "sum")).lower()
        table = {"sum": self._Add, "mul": self._Multiply, "ave": self._Average, "max": self._Maximum, "concat": self._Concatenate}
        if mode not in table:
            raise UnsupportedKerasConfigurationException("Keras Merge mode '%s' is not supported" % mode)
        c = dict(cfg)
        if mode == "concat":
            c.setdefault("axis", cfg.get("concat_axis", -1))
        return table[mode](c, s)

    # -- attention (KerasAttentionLayer -> DotProductAttentionLayer): inputs [query, value(, key)]
    def _Attention(self, cfg, s):
        layer = L.DotProductAttentionLayer(scale=1.0 if not cfg.get("use_scale", False) else None, dropoutProbability=float(cfg.get("dropout", 0.0)),
                                           useCausalMask=bool(cfg.get("causal", False)), name=cfg.get("name")) \
            if "scale" in getattr(L.DotProductAttentionLayer, "FIELDS", {}) else L.DotProductAttentionLayer(name=cfg.get("name"))
        return _Mapped(layer), s

    # -- normalisation
    def _BatchNormalization(self, cfg, s):
        axis = cfg.get("axis", -1)
        axis = axis[0] if isinstance(axis, (list, tuple)) else axis
        if axis not in (-1, len(s)):
            raise UnsupportedKerasConfigurationException("BatchNormalization over axis %s is not supported (channels axis only)" % axis)
        layer = L.BatchNormalization(eps=float(cfg.get("epsilon", 1e-3)), decay=float(cfg.get("momentum", 0.99)), useLogStd=False,
                                     lockGammaBeta=not (cfg.get("scale", True) or cfg.get("center", True)), name=cfg.get("name"))

        def conv(w, ctx):
            n = len(w["moving_mean"])
            return {"gamma": w.get("gamma", np.ones(n, np.float32)), "beta": w.get("beta", np.zeros(n, np.float32)),
                    "mean": w["moving_mean"], "var": w["moving_variance"]}
        return _Mapped(layer, weights=conv), s

    # -- recurrent
    def _LSTM(self, cfg, s):
        n = int(cfg["units"])
        layer = L.LSTM(nOut=n, activation=map_activation(cfg.get("activation", "tanh")),
                       gateActivationFn=map_activation(cfg.get("recurrent_activation", "sigmoid")),
                       forgetGateBiasInit=1.0 if cfg.get("unit_forget_bias", True) else 0.0, **self._common(cfg))
        seq = cfg.get("return_sequences", False)
        if not seq:
            layer = L.LastTimeStep(layer)

        def conv(w, ctx):
            def reorder(m):                                   # Keras [i, f, c, o] -> DL4J [c(a), f, o, i(g)]
                i, f, c, o = np.split(m, 4, axis=-1)
                return np.concatenate([c, f, o, i], axis=-1)
            out = {"W": reorder(w["kernel"]), "RW": reorder(w["recurrent_kernel"])}
            out["b"] = reorder(w["bias"]) if "bias" in w else np.zeros(4 * n, np.float32)
            return out
        return _Mapped(layer, weights=conv), ((s[0], n) if seq else (n,))

    def _SimpleRNN(self, cfg, s):
        n = int(cfg["units"])
        layer = L.SimpleRnn(nOut=n, activation=map_activation(cfg.get("activation", "tanh")), **self._common(cfg))
        seq = cfg.get("return_sequences", False)
        if not seq:
            layer = L.LastTimeStep(layer)
        return _Mapped(layer, weights=lambda w, ctx: {"W": w["kernel"], "RW": w["recurrent_kernel"], "b": w.get("bias", np.zeros(n, np.float32))}), \
            ((s[0], n) if seq else (n,))

    def _Bidirectional(self, cfg, s):
        inner = cfg["layer"]
        icfg = dict(inner["config"])
        seq = icfg.get("return_sequences", False)
        icfg["return_sequences"] = True
        m, out = self.map(inner["class_name"], icfg, s)
        mode = {"concat": "CONCAT", "sum": "ADD", "mul": "MUL", "ave": "AVERAGE"}[cfg.get("merge_mode", "concat")]
        layer = L.Bidirectional(m.layer, mode=mode, name=cfg.get("name"))
        n = out[-1] * (2 if mode == "CONCAT" else 1)
        if not seq:
            layer = L.LastTimeStep(layer)

        def conv(w, ctx):
            fw = m.weights({k[len("forward_"):]: v for k, v in w.items() if k.startswith("forward_")}, ctx)
            bw = m.weights({k[len("backward_"):]: v for k, v in w.items() if k.startswith("backward_")}, ctx)
            out = {"f" + k: v for k, v in fw.items()}
            out.update({"b" + k: v for k, v in bw.items()})
            return out
        return _Mapped(layer, weights=conv), ((s[0], n) if seq else (n,))

    def _TimeDistributed(self, cfg, s):
        inner = cfg["layer"]
        m, out = self.map(inner["class_name"], inner["config"], s[1:])
        layer = m.layer if isinstance(m.layer, L.TimeDistributed) else L.TimeDistributed(m.layer)
        return _Mapped(layer, weights=m.weights), (s[0],) + tuple(out)

    # -- merges (functional API)
    def _Add(self, cfg, s): return _Mapped(vertex=ElementWiseVertex("Add")), s
    def _Subtract(self, cfg, s): return _Mapped(vertex=ElementWiseVertex("Subtract")), s
    def _Multiply(self, cfg, s): return _Mapped(vertex=ElementWiseVertex("Product")), s
    def _Average(self, cfg, s): return _Mapped(vertex=ElementWiseVertex("Average")), s
    def _Maximum(self, cfg, s): return _Mapped(vertex=ElementWiseVertex("Max")), s

    def _Concatenate(self, cfg, s):
        return _Mapped(vertex=MergeVertex()), s


def _input_type(shape):
    shape = tuple(shape)
    if len(shape) == 3:
        return InputType.convolutional(shape[0], shape[1], shape[2])          # (h, w, c) -> NCHW network input
    if len(shape) == 2:
        return InputType.recurrent(shape[1], shape[0] if shape[0] else -1)
    if len(shape) == 4:
        return InputType.convolutional3D(shape[0], shape[1], shape[2], shape[3])
    return InputType.feedForward(shape[0])


def _load_weights(source) -> Dict[str, Dict[str, np.ndarray]]:
    """-> {layer_name: {weight_name: array}}"""
    if source is None:
        return {}
    if isinstance(source, dict):
        flat = source
    elif str(source).endswith(".npz"):
        flat = dict(np.load(source, allow_pickle=False))
    else:
        from .hdf5 import Hdf5Archive
        flat = Hdf5Archive(source).keras_weights()
    out: Dict[str, Dict[str, np.ndarray]] = {}
    for k, v in flat.items():
        if isinstance(v, dict):
            out[k] = {kk.split("/")[-1].split(":")[0]: np.asarray(a) for kk, a in v.items()}
            continue
        parts = k.split("/")
        layer, name = parts[0], parts[-1].split(":")[0]
        for comp in parts[1:-1]:                       # Bidirectional: <layer>/forward_lstm/.../kernel -> forward_kernel
            if comp.startswith("forward"):
                name = "forward_" + name
            elif comp.startswith("backward"):
                name = "backward_" + name
        out.setdefault(layer, {})[name] = np.asarray(v)
    return out


def _permute_dense_after_flatten(W: np.ndarray, shape3):
    """Rows of a Dense kernel that consumed a channels_last Flatten: (h, w, c) order -> (c, h, w) order."""
    h, w, c = shape3
    return W.reshape(h, w, c, -1).transpose(2, 0, 1, 3).reshape(h * w * c, -1)


def _set_params(net, layer_key, conv_w: Dict[str, np.ndarray]):
    table = net.paramTable()
    for pk, arr in conv_w.items():
        key = "%s_%s" % (layer_key, pk)
        if key not in table:
            raise InvalidKerasConfigurationException("Layer %s has no parameter %s (has: %s)" % (layer_key, pk,
                                                     [k for k in table if k.startswith(str(layer_key) + "_")]))
        t = table[key]
        a = torch.as_tensor(np.ascontiguousarray(arr), dtype=t.dtype)
        if a.numel() != t.numel():
            raise InvalidKerasConfigurationException("Shape mismatch for %s: keras %s vs dl4j %s" % (key, tuple(a.shape), tuple(t.shape)))
        t.copy_(a.reshape(t.shape))
    net._refresh_replicas()


class KerasModelImport:
    @staticmethod
    def _read_config(src):
        if isinstance(src, dict):
            return src
        if isinstance(src, str) and src.lstrip().startswith("{"):
            return json.loads(src)
        path = os.fspath(src)
        if path.endswith((".h5", ".hdf5", ".keras")):
            from .hdf5 import Hdf5Archive
            return json.loads(Hdf5Archive(path).read_attribute_as_string("model_config"))
        with open(path, encoding="utf-8") as f:
            return json.load(f)

    # ------------------------------------------------------------------ Sequential
    @staticmethod
    def importKerasSequentialConfiguration(modelJson, enforceTrainingConfig=False, loss=None, _return_plan=False):
        cfg = KerasModelImport._read_config(modelJson)
        if cfg.get("class_name") != "Sequential":
            raise InvalidKerasConfigurationException("Model class name must be Sequential (found %s)" % cfg.get("class_name"))
        layers = cfg["config"]["layers"] if isinstance(cfg["config"], dict) else cfg["config"]
        mapper = KerasLayerMapper(enforceTrainingConfig)
        shape = None
        first = layers[0]["config"]
        bis = first.get("batch_input_shape") or first.get("batch_shape")
        if bis:
            shape = tuple(bis[1:])
        elif isinstance(cfg["config"], dict) and cfg["config"].get("build_input_shape"):
            shape = tuple(cfg["config"]["build_input_shape"][1:])
        if shape is None:
            raise InvalidKerasConfigurationException("Sequential model has no input shape (batch_input_shape missing on the first layer)")
        in_type = _input_type(shape)
        lb = NeuralNetConfiguration.Builder().list()
        plan = []                # (dl4j layer index, keras layer name, mapped, pre-flatten shape or None)
        pending_flatten = None
        idx = 0
        for k, kl in enumerate(layers):
            m, shape_out = mapper.map(kl["class_name"], kl["config"], shape)
            if m.flatten:
                pending_flatten = tuple(shape) if len(shape) == 3 else None
            if m.layer is not None:
                layer = m.layer
                last = k == len(layers) - 1
                if last and isinstance(layer, L.DenseLayer):
                    lf = loss or ("MCXENT" if str(layer.activationFn).upper().find("SOFTMAX") >= 0 or type(layer.activationFn).__name__.endswith("Softmax") else "MSE")
                    layer = L.OutputLayer(nOut=layer.nOut, activation=layer.activationFn, lossFunction=lf, hasBias=layer.hasBias, name=layer.layerName,
                                          weightInit=layer.weightInitFn, l1=layer.l1, l2=layer.l2)
                lb.layer(idx, layer)
                plan.append((idx, kl["config"].get("name"), m, pending_flatten if isinstance(m.layer, L.DenseLayer) else None))
                if isinstance(m.layer, L.DenseLayer):
                    pending_flatten = None
                idx += 1
            shape = shape_out
        conf = lb.setInputType(in_type).build()
        return (conf, plan) if _return_plan else conf

    @staticmethod
    def importKerasSequentialModelAndWeights(modelJson, weights=None, enforceTrainingConfig=False, loss=None, device=None):
        from ..nn.multilayer import MultiLayerNetwork
        if weights is None and isinstance(modelJson, (str, os.PathLike)) and str(modelJson).endswith((".h5", ".hdf5")):
            weights = modelJson
        conf, plan = KerasModelImport.importKerasSequentialConfiguration(modelJson, enforceTrainingConfig, loss, _return_plan=True)
        net = MultiLayerNetwork(conf, device=device)
        net.init()
        W = _load_weights(weights)
        for idx, kname, m, pre_flat in plan:
            if m.weights is None:
                continue
            if kname not in W:
                if weights is not None:
                    raise InvalidKerasConfigurationException("No weights found for layer " + str(kname))
                continue
            w = dict(W[kname])
            if pre_flat is not None and "kernel" in w:
                w["kernel"] = _permute_dense_after_flatten(w["kernel"], pre_flat)
            _set_params(net, idx, m.weights(w, None))
        return net

    # ------------------------------------------------------------------ Functional
    @staticmethod
    def importKerasModelConfiguration(modelJson, enforceTrainingConfig=False, loss=None, _return_plan=False):
        cfg = KerasModelImport._read_config(modelJson)
        if cfg.get("class_name") not in ("Model", "Functional"):
            raise InvalidKerasConfigurationException("Model class name must be Model/Functional (found %s)" % cfg.get("class_name"))
        c = cfg["config"]
        mapper = KerasLayerMapper(enforceTrainingConfig)
        gb = NeuralNetConfiguration.Builder().graphBuilder()

        def names(spec):
            return [s[0] for s in spec] if spec and isinstance(spec[0], (list, tuple)) else [spec[0]]
        inputs = names(c["input_layers"])
        outputs = names(c["output_layers"])
        shapes: Dict[str, tuple] = {}
        alias: Dict[str, str] = {}          # keras layer name -> dl4j vertex that carries its output
        flat_from: Dict[str, tuple] = {}    # vertex output that is a flattened (h, w, c) tensor
        in_types = []
        plan = []
        gb.addInputs(*inputs)
        for kl in c["layers"]:
            name, cls, kcfg = kl["config"].get("name", kl.get("name")), kl["class_name"], kl["config"]
            inbound = kl.get("inbound_nodes") or []
            srcs = []
            if inbound:
                node = inbound[0]
                if isinstance(node, dict):                      # Keras 3: {"args": [...], "kwargs": {}}
                    def walk(a):
                        if isinstance(a, dict) and a.get("class_name") == "__keras_tensor__":
                            srcs.append(a["config"]["keras_history"][0])
                        elif isinstance(a, (list, tuple)):
                            for x in a:
                                walk(x)
                    walk(node.get("args", []))
                else:
                    srcs = [n[0] for n in node]
            if cls == "InputLayer":
                m, shp = mapper.map(cls, kcfg, None)
                shapes[name] = shp
                alias[name] = name
                in_types.append((name, _input_type(shp)))
                continue
            in_shape = shapes[srcs[0]]
            m, shp = mapper.map(cls, kcfg, in_shape)
            shapes[name] = shp
            dsrc = [alias[s] for s in srcs]
            if m.passthrough:
                alias[name] = dsrc[0]
                if m.flatten and len(in_shape) == 3:
                    flat_from[name] = tuple(in_shape)
                elif srcs[0] in flat_from and not m.flatten:
                    flat_from[name] = flat_from[srcs[0]]
                continue
            alias[name] = name
            if m.vertex is not None:
                gb.addVertex(name, m.vertex, *dsrc)
                continue
            layer = m.layer
            if name in outputs and isinstance(layer, L.DenseLayer):
                sm = type(layer.activationFn).__name__.lower().find("softmax") >= 0
                layer = L.OutputLayer(nOut=layer.nOut, activation=layer.activationFn, lossFunction=loss or ("MCXENT" if sm else "MSE"),
                                      hasBias=layer.hasBias, name=name, weightInit=layer.weightInitFn, l1=layer.l1, l2=layer.l2)
            gb.addLayer(name, layer, *dsrc)
            plan.append((name, name, m, flat_from.get(srcs[0]) if isinstance(m.layer, L.DenseLayer) else None))
        gb.setOutputs(*[alias[o] for o in outputs])
        order = {n: i for i, n in enumerate(inputs)}
        in_types.sort(key=lambda kv: order.get(kv[0], 0))
        gb.setInputTypes(*[t for _, t in in_types])
        conf = gb.build()
        return (conf, plan) if _return_plan else conf

    @staticmethod
    def importKerasModelAndWeights(modelJson, weights=None, enforceTrainingConfig=False, loss=None, device=None):
        from ..nn.graph.computation_graph import ComputationGraph
        cfg = KerasModelImport._read_config(modelJson)
        if cfg.get("class_name") == "Sequential":
            return KerasModelImport.importKerasSequentialModelAndWeights(cfg, weights if weights is not None else modelJson, enforceTrainingConfig, loss,
                                                                         device)
        if weights is None and isinstance(modelJson, (str, os.PathLike)) and str(modelJson).endswith((".h5", ".hdf5")):
            weights = modelJson
        conf, plan = KerasModelImport.importKerasModelConfiguration(cfg, enforceTrainingConfig, loss, _return_plan=True)
        net = ComputationGraph(conf, device=device)
        net.init()
        W = _load_weights(weights)
        for vname, kname, m, pre_flat in plan:
            if m.weights is None:
                continue
            if kname not in W:
                if weights is not None:
                    raise InvalidKerasConfigurationException("No weights found for layer " + str(kname))
                continue
            w = dict(W[kname])
            if pre_flat is not None and "kernel" in w:
                w["kernel"] = _permute_dense_after_flatten(w["kernel"], pre_flat)
            _set_params(net, vname, m.weights(w, None))
        return net


class KerasModel:
    """Functional-API Keras model (reference: modelimport/keras/KerasModel.java): parsed configuration + weights, materialised as a
    ComputationGraph. ``KerasModel.builder().modelJsonFilename(..).weightsHdf5Filename(..).enforceTrainingConfig(..).buildModel()``."""
    SEQUENTIAL = False

    def __init__(self, modelJson=None, weights=None, enforceTrainingConfig=False, loss=None):
        self.modelJson, self.weights = modelJson, weights
        self.enforceTrainingConfig, self.loss = enforceTrainingConfig, loss
        self._config = KerasModelImport._read_config(modelJson) if modelJson is not None else None

    class Builder:
        def __init__(self, cls):
            self._cls, self._kw = cls, {}

        def modelJson(self, s):
            self._kw["modelJson"] = s
            return self

        modelJsonFilename = modelJson
        modelHdf5Filename = modelJson

        def weightsHdf5Filename(self, p):
            self._kw["weights"] = p
            return self

        weightsHdf5FilenameNoRoot = weightsHdf5Filename

        def enforceTrainingConfig(self, b):
            self._kw["enforceTrainingConfig"] = bool(b)
            return self

        def buildModel(self):
            return KerasModel(**self._kw)

        def buildSequential(self):
            return KerasSequentialModel(**self._kw)

    @classmethod
    def builder(cls):
        return KerasModel.Builder(cls)

    def getClassName(self):
        return (self._config or {}).get("class_name")

    def getLayerNames(self):
        cfg = (self._config or {}).get("config", {})
        layers = cfg.get("layers", cfg) if isinstance(cfg, dict) else cfg
        return [l.get("config", {}).get("name") or l.get("name") for l in layers]

    def getComputationGraphConfiguration(self):
        return KerasModelImport.importKerasModelConfiguration(self._config, self.enforceTrainingConfig, self.loss)

    def getComputationGraph(self, importWeights=True, device=None):
        return KerasModelImport.importKerasModelAndWeights(self._config, self.weights if importWeights else None,
                                                           self.enforceTrainingConfig, self.loss, device=device)


class KerasSequentialModel(KerasModel):
    """Sequential Keras model -> MultiLayerNetwork (modelimport/keras/KerasSequentialModel.java)."""
    SEQUENTIAL = True

    def getMultiLayerConfiguration(self):
        return KerasModelImport.importKerasSequentialConfiguration(self._config, self.enforceTrainingConfig, self.loss)

    def getMultiLayerNetwork(self, importWeights=True, device=None):
        return KerasModelImport.importKerasSequentialModelAndWeights(self._config, self.weights if importWeights else None,
                                                                     self.enforceTrainingConfig, self.loss, device=device)
