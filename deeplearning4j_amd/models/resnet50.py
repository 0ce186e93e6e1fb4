"""Zoo ResNet-50 ComputationGraph, exactly as defined by the reference zoo
(deeplearning4j-zoo/.../zoo/model/ResNet50.java:52-59,94-238): RmsProp(0.1, 0.96, 0.001),
TruncatedNormal(0, 0.5) init, l1 1e-7, l2 5e-5, Truncate convolution mode (Same for the 3x3s),
ZeroPadding(3) -> 7x7/2 conv -> BN -> ReLU -> maxpool 3x3/2, stages [64,64,256]x3 (first block ALSO
stride 2), [128,128,512]x4, [256,256,1024]x6, [512,512,2048]x3, a layer *named* "avgpool" that is a
MAX 3x3 SubsamplingLayer, NLL/softmax OutputLayer. All convs have bias; BN/ReLU/Add are separate
graph nodes in the configuration (the runtime fusion planner merges them at execution time)."""
from __future__ import annotations

from ..nd4j.learning import RmsProp
from ..nn.conf.base import (InputType, ConvolutionMode, WorkspaceMode, CacheMode, OptimizationAlgorithm,
                            TruncatedNormalDistribution, WeightInitDistribution, PoolingType)
from ..nn.conf.configuration import NeuralNetConfiguration
from ..nn.conf.graph import ElementWiseVertex
from ..nn.conf.layers import (ConvolutionLayer, BatchNormalization, ActivationLayer, SubsamplingLayer, ZeroPaddingLayer,
                              OutputLayer)
from ..nn.graph.computation_graph import ComputationGraph
from .zoo_model import ZooModel


class ResNet50(ZooModel):
    def __init__(self, seed=1234, inputShape=(3, 224, 224), numClasses=0, weightInit=None, updater=None,
                 cacheMode=CacheMode.NONE, workspaceMode=WorkspaceMode.ENABLED, cudnnAlgoMode="PREFER_FASTEST",
                 dataType="FLOAT", computeDataType=None):
        self.seed, self.inputShape, self.numClasses = seed, tuple(inputShape), numClasses
        self.weightInit = weightInit or WeightInitDistribution(TruncatedNormalDistribution(0.0, 0.5))
        self.updater = updater or RmsProp(0.1, 0.96, 0.001)
        self.cacheMode, self.workspaceMode, self.cudnnAlgoMode = cacheMode, workspaceMode, cudnnAlgoMode
        self.dataType, self.computeDataType = dataType, computeDataType

    def _identity_block(self, g, k, f, stage, block, inp):
        cn, bn, an, sn = ("%s%s%s_branch" % (p, stage, block) for p in ("res", "bn", "act", "short"))
        (g.addLayer(cn + "2a", ConvolutionLayer(kernelSize=(1, 1), nOut=f[0], cudnnAlgoMode=self.cudnnAlgoMode), inp)
          .addLayer(bn + "2a", BatchNormalization(), cn + "2a")
          .addLayer(an + "2a", ActivationLayer(activation="RELU"), bn + "2a")
          .addLayer(cn + "2b", ConvolutionLayer(kernelSize=k, nOut=f[1], cudnnAlgoMode=self.cudnnAlgoMode,
                                                convolutionMode=ConvolutionMode.Same), an + "2a")
          .addLayer(bn + "2b", BatchNormalization(), cn + "2b")
          .addLayer(an + "2b", ActivationLayer(activation="RELU"), bn + "2b")
          .addLayer(cn + "2c", ConvolutionLayer(kernelSize=(1, 1), nOut=f[2], cudnnAlgoMode=self.cudnnAlgoMode), an + "2b")
          .addLayer(bn + "2c", BatchNormalization(), cn + "2c")
          .addVertex(sn, ElementWiseVertex(ElementWiseVertex.Op.Add), bn + "2c", inp)
          .addLayer(cn, ActivationLayer(activation="RELU"), sn))

    def _conv_block(self, g, k, f, stage, block, inp, stride=(2, 2)):
        cn, bn, an, sn = ("%s%s%s_branch" % (p, stage, block) for p in ("res", "bn", "act", "short"))
        (g.addLayer(cn + "2a", ConvolutionLayer(kernelSize=(1, 1), stride=stride, nOut=f[0]), inp)
          .addLayer(bn + "2a", BatchNormalization(), cn + "2a")
          .addLayer(an + "2a", ActivationLayer(activation="RELU"), bn + "2a")
          .addLayer(cn + "2b", ConvolutionLayer(kernelSize=k, nOut=f[1], convolutionMode=ConvolutionMode.Same), an + "2a")
          .addLayer(bn + "2b", BatchNormalization(), cn + "2b")
          .addLayer(an + "2b", ActivationLayer(activation="RELU"), bn + "2b")
          .addLayer(cn + "2c", ConvolutionLayer(kernelSize=(1, 1), nOut=f[2]), an + "2b")
          .addLayer(bn + "2c", BatchNormalization(), cn + "2c")
          .addLayer(cn + "1", ConvolutionLayer(kernelSize=(1, 1), stride=stride, nOut=f[2]), inp)
          .addLayer(bn + "1", BatchNormalization(), cn + "1")
          .addVertex(sn, ElementWiseVertex(ElementWiseVertex.Op.Add), bn + "2c", bn + "1")
          .addLayer(cn, ActivationLayer(activation="RELU"), sn))

    def graphBuilder(self):
        b = (NeuralNetConfiguration.Builder().seed(self.seed).activation("IDENTITY")
             .optimizationAlgo(OptimizationAlgorithm.STOCHASTIC_GRADIENT_DESCENT).updater(self.updater)
             .weightInit(self.weightInit).l1(1e-7).l2(5e-5).miniBatch(True).cacheMode(self.cacheMode)
             .trainingWorkspaceMode(self.workspaceMode).inferenceWorkspaceMode(self.workspaceMode)
             .cudnnAlgoMode(self.cudnnAlgoMode).convolutionMode(ConvolutionMode.Truncate)
             .dataType(self.dataType).computeDataType(self.computeDataType))
        g = b.graphBuilder()
        c, h, w = self.inputShape
        (g.addInputs("input").setInputTypes(InputType.convolutional(h, w, c))
          .addLayer("stem-zero", ZeroPaddingLayer(padding=(3, 3)), "input")
          .addLayer("stem-cnn1", ConvolutionLayer(kernelSize=(7, 7), stride=(2, 2), nOut=64), "stem-zero")
          .addLayer("stem-batch1", BatchNormalization(), "stem-cnn1")
          .addLayer("stem-act1", ActivationLayer(activation="RELU"), "stem-batch1")
          .addLayer("stem-maxpool1", SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(3, 3), stride=(2, 2)), "stem-act1"))
        k = (3, 3)
        self._conv_block(g, k, (64, 64, 256), "2", "a", "stem-maxpool1", (2, 2))
        self._identity_block(g, k, (64, 64, 256), "2", "b", "res2a_branch")
        self._identity_block(g, k, (64, 64, 256), "2", "c", "res2b_branch")
        self._conv_block(g, k, (128, 128, 512), "3", "a", "res2c_branch")
        for blk, prev in (("b", "a"), ("c", "b"), ("d", "c")):
            self._identity_block(g, k, (128, 128, 512), "3", blk, "res3%s_branch" % prev)
        self._conv_block(g, k, (256, 256, 1024), "4", "a", "res3d_branch")
        for blk, prev in (("b", "a"), ("c", "b"), ("d", "c"), ("e", "d"), ("f", "e")):
            self._identity_block(g, k, (256, 256, 1024), "4", blk, "res4%s_branch" % prev)
        self._conv_block(g, k, (512, 512, 2048), "5", "a", "res4f_branch")
        self._identity_block(g, k, (512, 512, 2048), "5", "b", "res5a_branch")
        self._identity_block(g, k, (512, 512, 2048), "5", "c", "res5b_branch")
        (g.addLayer("avgpool", SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(3, 3)), "res5c_branch")
          .addLayer("output", OutputLayer(lossFunction="NEGATIVELOGLIKELIHOOD", nOut=self.numClasses, activation="SOFTMAX"),
                    "avgpool")
          .setOutputs("output"))
        return g

    def conf(self):
        return self.graphBuilder().build()

    def init(self, device=None):
        net = ComputationGraph(self.conf(), device=device)
        net.init()
        return net
