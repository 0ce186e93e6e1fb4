"""Layer-helper SPI (reference: nn/layers/{LayerHelper, HelperUtils (:51-114)}.java and the ConvolutionHelper / SubsamplingHelper /
BatchNormalizationHelper / LSTMHelper / DropoutHelper interfaces that the cuDNN module implemented).

In the reference a layer asks HelperUtils for a vendor helper by class name and falls back to its built-in math. Here the
hand-written HIP kernels ARE the layers' GPU path (nn/layers/core.py -> ops/*.py -> native/csrc/*.hip); the helper objects below
answer the same questions (is a helper available for this layer on this device, what does it support) and are what
``layer.getHelper()`` returns, so helper-aware user code and tests have something real to talk to.
"""
from __future__ import annotations

import torch


class LayerHelper:
    NAME = "none"
    KERNELS = ()

    def __init__(self, device=None, dtype=None):
        self.device = torch.device(device) if device is not None else None
        self.dtype = dtype

    def checkSupported(self) -> bool:
        from ...native import lib as L
        if self.device is None or self.device.type != "cuda":
            return False
        try:
            L.hip()
            return True
        except Exception:
            return False

    def helperMemoryUse(self):
        return {}

    def __repr__(self):
        return "%s(%s, kernels=%s)" % (type(self).__name__, self.NAME, ", ".join(self.KERNELS))


class ConvolutionHelper(LayerHelper):
    NAME = "hip-igemm"
    KERNELS = ("igemm_fwd_kernel", "igemm_wgrad_kernel", "conv_c3_fwd_kernel", "conv_c3_wgrad_kernel")

    def supports(self, x, w, stride, padding, dilation):
        from ...ops import conv as CONV
        return CONV._hip_eligible(x, w, stride, padding, dilation) or CONV._stem_eligible(x, w, stride, padding, dilation)


class SubsamplingHelper(LayerHelper):
    NAME = "hip-pooling"
    KERNELS = ("pool_fwd_kernel", "pool_bwd_kernel", "pool3d_fwd_kernel", "pool3d_bwd_kernel")


class BatchNormalizationHelper(LayerHelper):
    NAME = "hip-batchnorm"
    KERNELS = ("bn_colreduce_kernel", "bn_finalize_partials_kernel", "bn_apply_kernel", "bn_bwd_dx_kernel")


class LSTMHelper(LayerHelper):
    NAME = "hip-lstm-cell"
    KERNELS = ("lstm_cell_fwd_kernel", "lstm_cell_bwd_kernel")


class DropoutHelper(LayerHelper):
    NAME = "hip-philox-dropout"
    KERNELS = ("dropout_kernel", "dropout_bwd_kernel")


class HelperUtils:
    """HelperUtils.createHelper(cudnnHelperClassName, oneDnnClassName, layerHelperSuperClass, layerName, args...)."""

    _BY_NAME = {"ConvolutionHelper": ConvolutionHelper, "SubsamplingHelper": SubsamplingHelper,
                "BatchNormalizationHelper": BatchNormalizationHelper, "LSTMHelper": LSTMHelper, "DropoutHelper": DropoutHelper}

    @staticmethod
    def createHelper(cudnnHelperClassName="", oneDnnClassName="", layerHelperSuperClass=LayerHelper, layerName="", *args, device=None,
                     dtype=None):
        from ...nd4j.backend import DL4JSystemProperties
        cls = layerHelperSuperClass if isinstance(layerHelperSuperClass, type) else HelperUtils._BY_NAME.get(str(layerHelperSuperClass))
        if cls is None or cls is LayerHelper:
            for k, v in HelperUtils._BY_NAME.items():
                if k.replace("Helper", "") in (cudnnHelperClassName or ""):
                    cls = v
        if cls is None or cls is LayerHelper:
            return None
        # reference default is "true" (DL4JSystemProperties.java:114-115 HELPER_DISABLE_DEFAULT_VALUE, HelperUtils.java:57):
        # the DL4J-level helper objects are opt-in there, the libnd4j platform helpers carry the acceleration. Same here:
        # the layer implementations call the HIP kernels directly; this object API only answers when explicitly enabled.
        if str(DL4JSystemProperties.get(DL4JSystemProperties.DISABLE_HELPER_PROPERTY,
                                        DL4JSystemProperties.HELPER_DISABLE_DEFAULT_VALUE)).lower() == "true":
            return None
        h = cls(device, dtype)
        return h if h.checkSupported() else None


def helper_for(layer_impl):
    """The helper object of a layer implementation instance (None on the CPU)."""
    from . import core, recurrent
    dev, dt = layer_impl.ctx.device, layer_impl.ctx.cdtype
    table = ((core.ConvolutionLayerImpl, ConvolutionHelper), (core.SubsamplingLayerImpl, SubsamplingHelper),
             (core.BatchNormalizationImpl, BatchNormalizationHelper), (recurrent.LSTMImpl, LSTMHelper))
    for impl_cls, h in table:
        if isinstance(layer_impl, impl_cls):
            obj = h(dev, dt)
            return obj if obj.checkSupported() else None
    return None
