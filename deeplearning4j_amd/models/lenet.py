"""Zoo LeNet (reference: deeplearning4j-zoo/.../zoo/model/LeNet.java): conv5x5x20 -> maxpool2 ->
conv5x5x50 -> maxpool2 -> dense500 relu -> softmax output, AdaDelta updater, XAVIER init."""
from __future__ import annotations

from ..nd4j.learning import AdaDelta
from ..nn.conf.base import InputType, ConvolutionMode, PoolingType
from ..nn.conf.configuration import NeuralNetConfiguration
from ..nn.conf.layers import ConvolutionLayer, SubsamplingLayer, DenseLayer, OutputLayer
from ..nn.multilayer import MultiLayerNetwork
from .zoo_model import ZooModel


class LeNet(ZooModel):
    def __init__(self, seed=1234, inputShape=(3, 224, 224), numClasses=0, updater=None, dataType="FLOAT",
                 computeDataType=None):
        self.seed, self.inputShape, self.numClasses = seed, tuple(inputShape), numClasses
        self.updater = updater or AdaDelta()
        self.dataType, self.computeDataType = dataType, computeDataType

    def conf(self):
        c, h, w = self.inputShape
        return (NeuralNetConfiguration.Builder().seed(self.seed).activation("IDENTITY").weightInit("XAVIER")
                .updater(self.updater).convolutionMode(ConvolutionMode.Same).dataType(self.dataType)
                .computeDataType(self.computeDataType)
                .list()
                .layer(0, ConvolutionLayer(name="cnn1", kernelSize=(5, 5), stride=(1, 1), nIn=c, nOut=20, activation="RELU"))
                .layer(1, SubsamplingLayer(name="maxpool1", poolingType=PoolingType.MAX, kernelSize=(2, 2), stride=(2, 2)))
                .layer(2, ConvolutionLayer(name="cnn2", kernelSize=(5, 5), stride=(1, 1), nOut=50, activation="RELU"))
                .layer(3, SubsamplingLayer(name="maxpool2", poolingType=PoolingType.MAX, kernelSize=(2, 2), stride=(2, 2)))
                .layer(4, DenseLayer(name="ffn1", activation="RELU", nOut=500))
                .layer(5, OutputLayer(name="output", lossFunction="MCXENT", nOut=self.numClasses, activation="SOFTMAX"))
                .setInputType(InputType.convolutionalFlat(h, w, c))
                .build())

    def init(self, device=None):
        net = MultiLayerNetwork(self.conf(), device=device)
        net.init()
        return net
