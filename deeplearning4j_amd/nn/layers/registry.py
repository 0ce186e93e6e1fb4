"""Layer config -> runtime implementation registry (DL4J: Layer.instantiate())."""
from __future__ import annotations

from ..conf import layers as C


def _table():
    from . import core, extra, recurrent as R
    return [
        (C.CenterLossOutputLayer, extra.CenterLossOutputLayerImpl),
        (C.OCNNOutputLayer, extra.OCNNOutputLayerImpl),
        (C.RnnOutputLayer, core.RnnOutputLayerImpl),
        (C.BaseOutputLayer, core.OutputLayerImpl),
        (C.LossLayer, core.LossLayerImpl),
        (C.Yolo2OutputLayer, extra.Yolo2OutputLayerImpl),
        (C.DenseLayer, core.DenseLayerImpl),
        (C.ActivationLayer, core.ActivationLayerImpl),
        (C.DropoutLayer, core.DropoutLayerImpl),
        (C.EmbeddingSequenceLayer, core.EmbeddingSequenceLayerImpl),
        (C.EmbeddingLayer, core.EmbeddingLayerImpl),
        (C.BatchNormalization, core.BatchNormalizationImpl),
        (C.LocalResponseNormalization, extra.LocalResponseNormalizationImpl),
        (C.PrimaryCapsules, extra.PrimaryCapsulesImpl),
        (C.LocallyConnected2D, extra.LocallyConnected2DImpl),
        (C.LocallyConnected1D, extra.LocallyConnected1DImpl),
        (C.Deconvolution3D, extra.Deconvolution3DImpl),
        (C.Convolution3D, extra.Convolution3DImpl),
        (C.Convolution1DLayer, extra.Convolution1DLayerImpl),
        (C.Deconvolution2D, extra.Deconvolution2DImpl),
        (C.DepthwiseConvolution2D, extra.DepthwiseConvolution2DImpl),
        (C.SeparableConvolution2D, extra.SeparableConvolution2DImpl),
        (C.ConvolutionLayer, core.ConvolutionLayerImpl),
        (C.Subsampling1DLayer, extra.Subsampling1DLayerImpl),
        (C.Subsampling3DLayer, extra.Subsampling3DLayerImpl),
        (C.SubsamplingLayer, core.SubsamplingLayerImpl),
        (C.Upsampling1D, extra.UpsamplingImpl), (C.Upsampling2D, extra.UpsamplingImpl), (C.Upsampling3D, extra.UpsamplingImpl),
        (C.ZeroPaddingLayer, core.ZeroPaddingLayerImpl),
        (C.ZeroPadding1DLayer, extra.ZeroPaddingNDImpl), (C.ZeroPadding3DLayer, extra.ZeroPaddingNDImpl),
        (C.Cropping1D, extra.CroppingImpl), (C.Cropping2D, extra.CroppingImpl), (C.Cropping3D, extra.CroppingImpl),
        (C.SpaceToDepthLayer, extra.SpaceToDepthImpl), (C.SpaceToBatchLayer, extra.SpaceToBatchImpl),
        (C.GlobalPoolingLayer, core.GlobalPoolingLayerImpl),
        (C.PReLULayer, extra.PReLULayerImpl),
        (C.ElementWiseMultiplicationLayer, extra.ElementWiseMultiplicationLayerImpl),
        (C.AutoEncoder, extra.AutoEncoderImpl),
        (C.VariationalAutoencoder, extra.VariationalAutoencoderImpl),
        (C.GravesBidirectionalLSTM, R.GravesBidirectionalLSTMImpl),
        (C.LSTM, R.LSTMImpl),
        (C.SimpleRnn, R.SimpleRnnImpl),
        (C.Bidirectional, R.BidirectionalImpl),
        (C.LastTimeStep, R.LastTimeStepImpl),
        (C.TimeDistributed, R.TimeDistributedImpl),
        (C.MaskZeroLayer, R.MaskZeroLayerImpl),
        (C.FrozenLayerWithBackprop, R.FrozenLayerWithBackpropImpl),
        (C.FrozenLayer, R.FrozenLayerImpl),
        (C.MaskLayer, R.MaskLayerImpl),
        (C.RepeatVector, R.RepeatVectorImpl),
        (C.LearnedSelfAttentionLayer, extra.LearnedSelfAttentionLayerImpl),
        (C.RecurrentAttentionLayer, extra.RecurrentAttentionLayerImpl),
        (C.SelfAttentionLayer, extra.SelfAttentionLayerImpl),
        (C.CapsuleLayer, extra.CapsuleLayerImpl),
        (C.CapsuleStrengthLayer, extra.CapsuleStrengthLayerImpl),
        (C.SameDiffLambdaLayer, extra.LambdaLayerImpl),
        (C.SameDiffOutputLayer, extra.SameDiffOutputLayerImpl),
        (C.SameDiffLayer, extra.SameDiffLayerImpl),
    ]


_T = None
_CUSTOM = []


def register(conf_cls, impl_cls):
    _CUSTOM.insert(0, (conf_cls, impl_cls))


def create(conf, ctx):
    global _T
    if _T is None:
        _T = _table()
    for cc, ic in _CUSTOM + _T:
        if isinstance(conf, cc):
            return ic(conf, ctx)
    raise ValueError("No runtime implementation registered for layer config %s" % type(conf).__name__)
