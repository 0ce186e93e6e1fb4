"""Graph fusion planner, CPU-checkable parts: ZeroPadding absorbed by the consuming convolution (identical numbers, no padded copy) and
the BatchNorm -> ReLU -> MaxPool node (falls back layer by layer without the HIP kernels)."""
import torch

from deeplearning4j_amd.nn.conf.base import InputType, PoolingType
from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
from deeplearning4j_amd.nn.conf.layers import (ActivationLayer, BatchNormalization, ConvolutionLayer, GlobalPoolingLayer, OutputLayer,
                                               SubsamplingLayer, ZeroPaddingLayer)
from deeplearning4j_amd.nn.graph.computation_graph import ComputationGraph, _FusedBNPoolNode
from deeplearning4j_amd.nd4j.learning import Sgd


def _net(fuse, asym=False):
    g = (NeuralNetConfiguration.Builder().seed(3).updater(Sgd(0.05)).weightInit("XAVIER").graphBuilder().addInputs("in")
         .addLayer("zp", ZeroPaddingLayer(padding=(2, 3, 2, 2) if asym else (2, 2)), "in")
         .addLayer("c1", ConvolutionLayer(kernelSize=(5, 5), stride=(2, 2), nOut=8, activation="IDENTITY"), "zp")
         .addLayer("bn", BatchNormalization(), "c1")
         .addLayer("act", ActivationLayer(activation="RELU"), "bn")
         .addLayer("pool", SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(3, 3), stride=(2, 2)), "act")
         .addLayer("gp", GlobalPoolingLayer(poolingType=PoolingType.AVG), "pool")
         .addLayer("out", OutputLayer(nOut=4, activation="SOFTMAX", lossFunction="MCXENT"), "gp")
         .setOutputs("out").setInputTypes(InputType.convolutional(15, 15, 3)).build())
    n = ComputationGraph(g)
    n.fusion_enabled = fuse
    n.init()
    return n


def test_zero_padding_is_absorbed_by_the_convolution_and_changes_nothing():
    a, b = _net(True), _net(False)
    b.setParams(a.params().clone())
    plan = a._build_plan()
    assert a.getLayer("zp").passthrough and a.getLayer("c1").extra_pad == (2, 2)
    assert any(isinstance(n, _FusedBNPoolNode) for n in plan)
    assert not b.getLayer("zp").passthrough
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 3, 15, 15, generator=g)
    y = torch.eye(4)[torch.randint(0, 4, (5,), generator=g)]
    a.computeGradientAndScore([x], [y])
    b.computeGradientAndScore([x], [y])
    assert abs(float(a.score()) - float(b.score())) < 1e-6
    assert float((a._grads - b._grads).abs().max()) < 1e-6
    assert float((a.outputSingle(x) - b.outputSingle(x)).abs().max()) < 1e-6
    # activations of every vertex are still available (the planner is switched off for feedForward)
    acts = a.feedForward([x], train=False)
    assert tuple(acts["zp"].shape) == (5, 3, 19, 19)


def test_asymmetric_padding_is_left_alone():
    a = _net(True, asym=True)
    a._build_plan()
    assert not a.getLayer("zp").passthrough and a.getLayer("c1").extra_pad == (0, 0)
