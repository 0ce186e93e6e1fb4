"""Model zoo beyond ResNet50/LeNet (reference: deeplearning4j-zoo/src/main/java/org/deeplearning4j/zoo/model/*.java and
model/helper/{DarknetHelper,FaceNetHelper,InceptionResNetHelper,NASNetHelper}.java).

Every model reproduces the reference's layer names, hyper-parameters and graph topology so that a DL4J checkpoint of
the same architecture lines up parameter-for-parameter.  Pretrained weights are not bundled (no network), see
``ZooModel.initPretrained(path)``.
"""
from __future__ import annotations

from ..nd4j.learning import AdaDelta, Adam, Nesterovs, RmsProp
from ..nn.conf.base import (CacheMode, ConvolutionMode, GradientNormalization, InputType, NormalDistribution,
                            OptimizationAlgorithm, PoolingType, TruncatedNormalDistribution, WeightInitDistribution,
                            WorkspaceMode, BackpropType)
from ..nn.conf.configuration import NeuralNetConfiguration
from ..nn.conf.graph import ElementWiseVertex, L2NormalizeVertex, MergeVertex, ScaleVertex
from ..nn.conf.layers import (ActivationLayer, BatchNormalization, CenterLossOutputLayer, CnnLossLayer, ConvolutionLayer,
                              Cropping2D, DenseLayer, DropoutLayer, GlobalPoolingLayer, GravesLSTM,
                              LocalResponseNormalization, LossLayer, OutputLayer, RnnOutputLayer, SeparableConvolution2D,
                              SpaceToDepthLayer, SubsamplingLayer, Upsampling2D, Yolo2OutputLayer, ZeroPaddingLayer)
from ..nn.graph.computation_graph import ComputationGraph
from ..nn.multilayer import MultiLayerNetwork
from .zoo_model import ZooModel

Add = ElementWiseVertex.Op.Add


class _Zoo(ZooModel):
    DEFAULT_SHAPE = (3, 224, 224)
    GRAPH = True

    def __init__(self, seed=1234, inputShape=None, numClasses=0, updater=None, weightInit=None, cacheMode=CacheMode.NONE,
                 workspaceMode=WorkspaceMode.ENABLED, cudnnAlgoMode="PREFER_FASTEST", dataType="FLOAT", computeDataType=None, **extra):
        self.seed, self.numClasses = seed, numClasses
        self.inputShape = tuple(inputShape or self.DEFAULT_SHAPE)
        self.updater = updater or self._default_updater()
        self.weightInit = weightInit
        self.cacheMode, self.workspaceMode, self.cudnnAlgoMode = cacheMode, workspaceMode, cudnnAlgoMode
        self.dataType, self.computeDataType = dataType, computeDataType
        for k, v in extra.items():
            if not hasattr(self, k):
                raise TypeError("%s: unknown option '%s'" % (type(self).__name__, k))
            setattr(self, k, v)

    def _default_updater(self):
        return Nesterovs(1e-2, 0.9)

    def _base(self):
        return (NeuralNetConfiguration.Builder().seed(self.seed).optimizationAlgo(OptimizationAlgorithm.STOCHASTIC_GRADIENT_DESCENT)
                .updater(self.updater).cacheMode(self.cacheMode).trainingWorkspaceMode(self.workspaceMode)
                .inferenceWorkspaceMode(self.workspaceMode).dataType(self.dataType).computeDataType(self.computeDataType))

    def _cnn_input(self):
        c, h, w = self.inputShape
        return InputType.convolutional(h, w, c)

    def init(self, device=None):
        net = (ComputationGraph if self.GRAPH else MultiLayerNetwork)(self.conf(), device=device)
        net.init()
        return net

    def modelType(self):
        return ComputationGraph if self.GRAPH else MultiLayerNetwork


def _conv(k, nOut, stride=(1, 1), **kw):
    k = (k, k) if isinstance(k, int) else tuple(k)
    stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
    return ConvolutionLayer(kernelSize=k, stride=stride, nOut=nOut, **kw)


def _act(a="RELU"):
    return ActivationLayer(activation=a)


def _pool(kind, k, s=None, **kw):
    k = (k, k) if isinstance(k, int) else tuple(k)
    s = k if s is None else ((s, s) if isinstance(s, int) else tuple(s))
    return SubsamplingLayer(poolingType=kind, kernelSize=k, stride=s, **kw)


# ============================================================================= AlexNet
class AlexNet(_Zoo):
    """AlexNet.java:71-165 (one-tower variant with LRN, biasUpdater Nesterovs(2e-2), RenormalizeL2PerLayer)."""
    GRAPH = False

    def _default_updater(self):
        return Nesterovs(1e-2, 0.9)

    def conf(self):
        c, h, w = self.inputShape
        nz = 1.0
        d005 = WeightInitDistribution(NormalDistribution(0.0, 0.005))
        return (self._base().weightInit(WeightInitDistribution(NormalDistribution(0.0, 0.01))).activation("RELU")
                .biasUpdater(Nesterovs(2e-2, 0.9)).convolutionMode(ConvolutionMode.Same)
                .gradientNormalization(GradientNormalization.RenormalizeL2PerLayer).l2(5e-4).miniBatch(False).list()
                .layer(0, _conv(11, 96, 4, name="cnn1", convolutionMode=ConvolutionMode.Truncate, nIn=c))
                .layer(1, LocalResponseNormalization())
                .layer(2, _pool(PoolingType.MAX, 3, 2, padding=(1, 1), name="maxpool1"))
                .layer(3, _conv(5, 256, 1, padding=(2, 2), name="cnn2", convolutionMode=ConvolutionMode.Truncate, biasInit=nz))
                .layer(4, _pool(PoolingType.MAX, 3, 2, convolutionMode=ConvolutionMode.Truncate, name="maxpool2"))
                .layer(5, LocalResponseNormalization())
                .layer(6, _conv(3, 384, 1, convolutionMode=ConvolutionMode.Same, name="cnn3"))
                .layer(7, _conv(3, 384, 1, name="cnn4", biasInit=nz))
                .layer(8, _conv(3, 256, 1, name="cnn5", biasInit=nz))
                .layer(9, _pool(PoolingType.MAX, 3, 2, name="maxpool3", convolutionMode=ConvolutionMode.Truncate))
                .layer(10, DenseLayer(name="ffn1", nOut=4096, weightInit=d005, biasInit=nz))
                .layer(11, DenseLayer(name="ffn2", nOut=4096, weightInit=d005, biasInit=nz, dropOut=0.5))
                .layer(12, OutputLayer(name="output", lossFunction="NEGATIVELOGLIKELIHOOD", nOut=self.numClasses, activation="SOFTMAX",
                                       weightInit=d005, biasInit=0.1))
                .setInputType(InputType.convolutional(h, w, c)).build())


# ============================================================================= SimpleCNN
class SimpleCNN(_Zoo):
    """SimpleCNN.java:70-138."""
    GRAPH = False
    DEFAULT_SHAPE = (3, 48, 48)

    def _default_updater(self):
        return AdaDelta()

    def conf(self):
        c, h, w = self.inputShape
        lb = (self._base().activation("IDENTITY").weightInit("RELU").convolutionMode(ConvolutionMode.Same).list())
        i = 0

        def add(l):
            nonlocal i
            lb.layer(i, l)
            i += 1
        add(_conv(7, 16, name="image_array", nIn=c)); add(BatchNormalization()); add(_conv(7, 16)); add(BatchNormalization())
        add(_act()); add(_pool(PoolingType.AVG, 2, 2)); add(DropoutLayer(dropOut=0.5))
        for k, n in ((5, 32), (3, 64), (3, 128)):
            add(_conv(k, n)); add(BatchNormalization()); add(_conv(k, n)); add(BatchNormalization())
            add(_act()); add(_pool(PoolingType.AVG, 2, 2)); add(DropoutLayer(dropOut=0.5))
        add(_conv(3, 256)); add(BatchNormalization()); add(_conv(3, self.numClasses))
        add(GlobalPoolingLayer(poolingType=PoolingType.AVG)); add(_act("SOFTMAX"))
        return lb.setInputType(InputType.convolutional(h, w, c)).build()


# ============================================================================= VGG16 / VGG19
class _VGG(_Zoo):
    BLOCKS = ()

    def _default_updater(self):
        return Nesterovs()

    def conf(self):
        c, h, w = self.inputShape
        g = self._base().activation("RELU").graphBuilder().addInputs("in")
        idx, prev = 0, "in"
        for nconv, nout in self.BLOCKS:
            for _ in range(nconv):
                g.addLayer(str(idx), _conv(3, nout, 1, padding=(1, 1), cudnnAlgoMode=self.cudnnAlgoMode, **({"nIn": c} if idx == 0 else {})), prev)
                prev, idx = str(idx), idx + 1
            g.addLayer(str(idx), _pool(PoolingType.MAX, 2, 2), prev)
            prev, idx = str(idx), idx + 1
        g.addLayer(str(idx), DenseLayer(nOut=4096, dropOut=0.5), prev)
        g.addLayer(str(idx + 1), DenseLayer(nOut=4096, dropOut=0.5), str(idx))
        g.addLayer(str(idx + 2), OutputLayer(name="output", lossFunction="NEGATIVELOGLIKELIHOOD", nOut=self.numClasses, activation="SOFTMAX"), str(idx + 1))
        return g.setOutputs(str(idx + 2)).setInputTypes(InputType.convolutionalFlat(h, w, c)).build()


class VGG16(_VGG):
    """VGG16.java:90-160 (layers named "0".."20")."""
    BLOCKS = ((2, 64), (2, 128), (3, 256), (3, 512), (3, 512))


class VGG19(_VGG):
    """VGG19.java (layers named "0".."23")."""
    BLOCKS = ((2, 64), (2, 128), (4, 256), (4, 512), (4, 512))


# ============================================================================= Darknet family
def _darknet_add(g, n, k, nIn, nOut, pool, poolStride=None, inp=None):
    """DarknetHelper.addLayers: conv(no bias, Same) -> BN -> LeakyReLU(0.1) [-> maxpool Same]."""
    if inp is None:
        inp = next((c for c in ("maxpooling2d_%d" % (n - 1), "activation_%d" % (n - 1), "concatenate_%d" % (n - 1)) if c in g.getVertices()), "input")
    (g.addLayer("convolution2d_%d" % n, _conv(k, nOut, 1, nIn=nIn, weightInit="XAVIER", convolutionMode=ConvolutionMode.Same, hasBias=False,
                                               activation="IDENTITY"), inp)
      .addLayer("batchnormalization_%d" % n, BatchNormalization(nIn=nOut, nOut=nOut, weightInit="XAVIER", activation="IDENTITY"), "convolution2d_%d" % n)
      .addLayer("activation_%d" % n, ActivationLayer(activation={"@class": "org.nd4j.linalg.activations.impl.ActivationLReLU", "alpha": 0.1}),
                "batchnormalization_%d" % n))
    if pool > 0:
        ps = pool if poolStride is None else poolStride
        g.addLayer("maxpooling2d_%d" % n, SubsamplingLayer(kernelSize=(pool, pool), stride=(ps, ps), convolutionMode=ConvolutionMode.Same), "activation_%d" % n)
    return g


_DARKNET19_BODY = ((3, 32, 2), (3, 64, 2), (3, 128, 0), (1, 64, 0), (3, 128, 2), (3, 256, 0), (1, 128, 0), (3, 256, 2), (3, 512, 0), (1, 256, 0),
                   (3, 512, 0), (1, 256, 0), (3, 512, 2), (3, 1024, 0), (1, 512, 0), (3, 1024, 0), (1, 512, 0), (3, 1024, 0))


class Darknet19(_Zoo):
    """Darknet19.java:87-149."""

    def _default_updater(self):
        return Nesterovs(1e-3, 0.9)

    def conf(self):
        g = (self._base().weightInit(self.weightInit or "RELU").l2(1e-5).activation("IDENTITY").cudnnAlgoMode(self.cudnnAlgoMode)
             .graphBuilder().addInputs("input").setInputTypes(self._cnn_input()))
        nin = self.inputShape[0]
        for n, (k, nout, pool) in enumerate(_DARKNET19_BODY, 1):
            _darknet_add(g, n, k, nin, nout, pool)
            nin = nout
        (g.addLayer("convolution2d_19", _conv(1, self.numClasses, 1, nIn=1024, convolutionMode=ConvolutionMode.Same, weightInit="RELU",
                                              activation="IDENTITY"), "activation_18")
          .addLayer("globalpooling", GlobalPoolingLayer(poolingType=PoolingType.AVG), "convolution2d_19")
          .addLayer("softmax", _act("SOFTMAX"), "globalpooling")
          .addLayer("loss", LossLayer(lossFunction="NEGATIVELOGLIKELIHOOD"), "softmax").setOutputs("loss"))
        return g.build()


class TinyYOLO(_Zoo):
    """TinyYOLO.java:87-142."""
    DEFAULT_SHAPE = (3, 416, 416)
    nBoxes = 5
    priorBoxes = ((1.08, 1.19), (3.42, 4.41), (6.63, 11.38), (9.42, 5.11), (16.62, 10.52))

    def _default_updater(self):
        return Adam(1e-3)

    def _yolo_base(self):
        return (self._base().gradientNormalization(GradientNormalization.RenormalizeL2PerLayer).gradientNormalizationThreshold(1.0)
                .l2(1e-5).activation("IDENTITY").cudnnAlgoMode(self.cudnnAlgoMode).graphBuilder().addInputs("input")
                .setInputTypes(self._cnn_input()))

    def conf(self):
        g = self._yolo_base()
        nin = self.inputShape[0]
        for n, (nout, pool, ps) in enumerate(((16, 2, 2), (32, 2, 2), (64, 2, 2), (128, 2, 2), (256, 2, 2), (512, 2, 1), (1024, 0, 0), (1024, 0, 0)), 1):
            _darknet_add(g, n, 3, nin, nout, pool, ps)
            nin = nout
        (g.addLayer("convolution2d_9", _conv(1, self.nBoxes * (5 + self.numClasses), 1, nIn=1024, convolutionMode=ConvolutionMode.Same,
                                             weightInit="RELU", activation="IDENTITY"), "activation_8")
          .addLayer("outputs", Yolo2OutputLayer(boundingBoxes=[list(p) for p in self.priorBoxes]), "convolution2d_9").setOutputs("outputs"))
        return g.build()


class YOLO2(TinyYOLO):
    """YOLO2.java:94-174 (Darknet19 body + 2 extra 3x3x1024 + passthrough reorg route)."""
    DEFAULT_SHAPE = (3, 608, 608)
    priorBoxes = ((0.57273, 0.677385), (1.87446, 2.06253), (3.33843, 5.47434), (7.88282, 3.52778), (9.77052, 9.16828))

    def conf(self):
        g = self._yolo_base()
        nin = self.inputShape[0]
        for n, (k, nout, pool) in enumerate(_DARKNET19_BODY, 1):
            _darknet_add(g, n, k, nin, nout, pool)
            nin = nout
        _darknet_add(g, 19, 3, 1024, 1024, 0)
        _darknet_add(g, 20, 3, 1024, 1024, 0)
        _darknet_add(g, 21, 1, 512, 64, 0, 0, inp="activation_13")
        g.addLayer("rearrange_21", SpaceToDepthLayer(blockSize=2), "activation_21")
        g.addVertex("concatenate_21", MergeVertex(), "rearrange_21", "activation_20")
        _darknet_add(g, 22, 3, 1024 + 256, 1024, 0, 0, inp="concatenate_21")
        (g.addLayer("convolution2d_23", _conv(1, self.nBoxes * (5 + self.numClasses), 1, nIn=1024, convolutionMode=ConvolutionMode.Same,
                                              weightInit="RELU", activation="IDENTITY", cudnnAlgoMode=self.cudnnAlgoMode), "activation_22")
          .addLayer("outputs", Yolo2OutputLayer(boundingBoxes=[list(p) for p in self.priorBoxes]), "convolution2d_23").setOutputs("outputs"))
        return g.build()


# ============================================================================= SqueezeNet
class SqueezeNet(_Zoo):
    """SqueezeNet.java:104-178 (v1.1 fire modules)."""
    DEFAULT_SHAPE = (3, 227, 227)

    def _default_updater(self):
        return Adam(1e-3)

    def _fire(self, g, fid, sq, ex, inp):
        p = "fire%d" % fid
        (g.addLayer(p + "_sq1x1", _conv(1, sq, cudnnAlgoMode=self.cudnnAlgoMode), inp)
          .addLayer(p + "_relu_sq1x1", _act(), p + "_sq1x1")
          .addLayer(p + "_exp1x1", _conv(1, ex, cudnnAlgoMode=self.cudnnAlgoMode), p + "_relu_sq1x1")
          .addLayer(p + "_relu_exp1x1", _act(), p + "_exp1x1")
          .addLayer(p + "_exp3x3", _conv(3, ex, convolutionMode=ConvolutionMode.Same, cudnnAlgoMode=self.cudnnAlgoMode), p + "_relu_sq1x1")
          .addLayer(p + "_relu_exp3x3", _act(), p + "_exp3x3")
          .addVertex(p, MergeVertex(), p + "_relu_exp1x1", p + "_relu_exp3x3"))
        return p

    def graphBuilder(self):
        g = (self._base().weightInit(self.weightInit or "RELU").l2(5e-5).miniBatch(True).convolutionMode(ConvolutionMode.Truncate).graphBuilder())
        (g.addLayer("conv1", _conv(3, 64, 2, cudnnAlgoMode=self.cudnnAlgoMode), "input").addLayer("conv1_act", _act(), "conv1")
          .addLayer("pool1", _pool(PoolingType.MAX, 3, 2), "conv1_act"))
        self._fire(g, 2, 16, 64, "pool1"); self._fire(g, 3, 16, 64, "fire2")
        g.addLayer("pool3", _pool(PoolingType.MAX, 3, 2), "fire3")
        self._fire(g, 4, 32, 128, "pool3"); self._fire(g, 5, 32, 128, "fire4")
        g.addLayer("pool5", _pool(PoolingType.MAX, 3, 2), "fire5")
        self._fire(g, 6, 48, 192, "pool5"); self._fire(g, 7, 48, 192, "fire6"); self._fire(g, 8, 64, 256, "fire7"); self._fire(g, 9, 64, 256, "fire8")
        (g.addLayer("drop9", DropoutLayer(dropOut=0.5), "fire9")
          .addLayer("conv10", _conv(1, self.numClasses, cudnnAlgoMode=self.cudnnAlgoMode), "drop9")
          .addLayer("conv10_act", _act(), "conv10").addLayer("avg_pool", GlobalPoolingLayer(poolingType=PoolingType.AVG), "conv10_act")
          .addLayer("softmax", _act("SOFTMAX"), "avg_pool").addLayer("loss", LossLayer(lossFunction="MCXENT"), "softmax").setOutputs("loss"))
        return g

    def conf(self):
        return self.graphBuilder().addInputs("input").setInputTypes(self._cnn_input()).build()


# ============================================================================= UNet
class UNet(_Zoo):
    """UNet.java:93-217 (sigmoid/XENT CnnLossLayer segmentation head)."""
    DEFAULT_SHAPE = (3, 512, 512)

    def _default_updater(self):
        return AdaDelta()

    def graphBuilder(self):
        g = self._base().weightInit(self.weightInit or "RELU").l2(5e-5).miniBatch(True).graphBuilder()

        def c(name, k, n, inp, act="RELU"):
            g.addLayer(name, _conv(k, n, 1, convolutionMode=ConvolutionMode.Same, cudnnAlgoMode=self.cudnnAlgoMode, activation=act), inp)
        prev = "input"
        for lvl, n in ((1, 64), (2, 128), (3, 256)):
            c("conv%d-1" % lvl, 3, n, prev); c("conv%d-2" % lvl, 3, n, "conv%d-1" % lvl)
            g.addLayer("pool%d" % lvl, SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(2, 2)), "conv%d-2" % lvl)
            prev = "pool%d" % lvl
        c("conv4-1", 3, 512, "pool3"); c("conv4-2", 3, 512, "conv4-1")
        g.addLayer("drop4", DropoutLayer(dropOut=0.5), "conv4-2")
        g.addLayer("pool4", SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(2, 2)), "drop4")
        c("conv5-1", 3, 1024, "pool4"); c("conv5-2", 3, 1024, "conv5-1")
        g.addLayer("drop5", DropoutLayer(dropOut=0.5), "conv5-2")
        prev = "drop5"
        for lvl, n, skip in ((6, 512, "drop4"), (7, 256, "conv3-2"), (8, 128, "conv2-2"), (9, 64, "conv1-2")):
            g.addLayer("up%d-1" % lvl, Upsampling2D(size=(2, 2)), prev)
            c("up%d-2" % lvl, 2, n, "up%d-1" % lvl)
            g.addVertex("merge%d" % lvl, MergeVertex(), skip, "up%d-2" % lvl)
            c("conv%d-1" % lvl, 3, n, "merge%d" % lvl); c("conv%d-2" % lvl, 3, n, "conv%d-1" % lvl)
            prev = "conv%d-2" % lvl
        c("conv9-3", 3, 2, "conv9-2")
        c("conv10", 1, 1, "conv9-3", act="IDENTITY")
        g.addLayer("output", CnnLossLayer(lossFunction="XENT", activation="SIGMOID"), "conv10").setOutputs("output")
        return g

    def conf(self):
        return self.graphBuilder().addInputs("input").setInputTypes(self._cnn_input()).build()


# ============================================================================= Xception
class Xception(_Zoo):
    """Xception.java:95-241."""
    DEFAULT_SHAPE = (3, 299, 299)

    def _default_updater(self):
        return AdaDelta()

    def graphBuilder(self):
        g = (self._base().weightInit(self.weightInit or "RELU").l2(4e-5).miniBatch(True).convolutionMode(ConvolutionMode.Truncate).graphBuilder())
        am = self.cudnnAlgoMode

        def sep(name, n, inp):
            g.addLayer(name, SeparableConvolution2D(kernelSize=(3, 3), nOut=n, hasBias=False, convolutionMode=ConvolutionMode.Same, cudnnAlgoMode=am), inp)
            g.addLayer(name + "_bn", BatchNormalization(), name)

        def residual(name, n, inp):
            g.addLayer(name + "_conv", _conv(1, n, 2, hasBias=False, convolutionMode=ConvolutionMode.Same, cudnnAlgoMode=am), inp)
            g.addLayer(name, BatchNormalization(), name + "_conv")

        def spool(name, inp):
            g.addLayer(name, _pool(PoolingType.MAX, 3, 2, convolutionMode=ConvolutionMode.Same), inp)
        (g.addLayer("block1_conv1", _conv(3, 32, 2, hasBias=False, cudnnAlgoMode=am), "input").addLayer("block1_conv1_bn", BatchNormalization(), "block1_conv1")
          .addLayer("block1_conv1_act", _act(), "block1_conv1_bn")
          .addLayer("block1_conv2", _conv(3, 64, 1, hasBias=False, cudnnAlgoMode=am), "block1_conv1_act")
          .addLayer("block1_conv2_bn", BatchNormalization(), "block1_conv2").addLayer("block1_conv2_act", _act(), "block1_conv2_bn"))
        residual("residual1", 128, "block1_conv2_act")
        sep("block2_sepconv1", 128, "block1_conv2_act")
        g.addLayer("block2_sepconv1_act", _act(), "block2_sepconv1_bn")
        sep("block2_sepconv2", 128, "block2_sepconv1_act")
        spool("block2_pool", "block2_sepconv2_bn")
        g.addVertex("add1", ElementWiseVertex(Add), "block2_pool", "residual1")
        for blk, n, add_in, add_out in ((3, 256, "add1", "add2"), (4, 728, "add2", "add3")):
            residual("residual%d" % (blk - 1), n, add_in)
            b = "block%d" % blk
            g.addLayer(b + "_sepconv1_act", _act(), add_in)
            sep(b + "_sepconv1", n, b + "_sepconv1_act")
            g.addLayer(b + "_sepconv2_act", _act(), b + "_sepconv1_bn")
            sep(b + "_sepconv2", n, b + "_sepconv2_act")
            spool(b + "_pool", b + "_sepconv2_bn")
            g.addVertex(add_out, ElementWiseVertex(Add), b + "_pool", "residual%d" % (blk - 1))
        res, blk = 3, 5
        for _ in range(8):
            prev, b = "add%d" % res, "block%d" % blk
            for j in (1, 2, 3):
                g.addLayer("%s_sepconv%d_act" % (b, j), _act(), prev if j == 1 else "%s_sepconv%d_bn" % (b, j - 1))
                sep("%s_sepconv%d" % (b, j), 728, "%s_sepconv%d_act" % (b, j))
            g.addVertex("add%d" % (res + 1), ElementWiseVertex(Add), b + "_sepconv3_bn", prev)
            res, blk = res + 1, blk + 1
        residual("residual12", 1024, "add%d" % res)
        g.addLayer("block13_sepconv1_act", _act(), "add11")
        sep("block13_sepconv1", 728, "block13_sepconv1_act")
        g.addLayer("block13_sepconv2_act", _act(), "block13_sepconv1_bn")
        sep("block13_sepconv2", 1024, "block13_sepconv2_act")
        spool("block13_pool", "block13_sepconv2_bn")
        g.addVertex("add12", ElementWiseVertex(Add), "block13_pool", "residual12")
        sep("block14_sepconv1", 1536, "add12")
        g.addLayer("block14_sepconv1_act", _act(), "block14_sepconv1_bn")
        sep("block14_sepconv2", 2048, "block14_sepconv1_act")
        (g.addLayer("block14_sepconv2_act", _act(), "block14_sepconv2_bn")
          .addLayer("avg_pool", GlobalPoolingLayer(poolingType=PoolingType.AVG), "block14_sepconv2_act")
          .addLayer("predictions", OutputLayer(lossFunction="MCXENT", nOut=self.numClasses, activation="SOFTMAX"), "avg_pool").setOutputs("predictions"))
        return g

    def conf(self):
        return self.graphBuilder().addInputs("input").setInputTypes(self._cnn_input()).build()


# ============================================================================= TextGenerationLSTM
class TextGenerationLSTM(_Zoo):
    """TextGenerationLSTM.java:73-93: 2x GravesLSTM(256, tanh) + RnnOutput(MCXENT/softmax), tBPTT 50/50."""
    GRAPH = False
    maxLength = 40
    totalUniqueCharacters = 47

    def __init__(self, **kw):
        self.maxLength = kw.pop("maxLength", 40)
        self.totalUniqueCharacters = kw.pop("totalUniqueCharacters", 47)
        kw.setdefault("inputShape", (self.maxLength, self.totalUniqueCharacters))
        super().__init__(**kw)

    def _default_updater(self):
        return RmsProp(0.01)

    def conf(self):
        return (self._base().seed(12345).l2(0.001).weightInit("XAVIER").cudnnAlgoMode(self.cudnnAlgoMode).list()
                .layer(0, GravesLSTM(nIn=self.inputShape[1], nOut=256, activation="TANH"))
                .layer(1, GravesLSTM(nOut=256, activation="TANH"))
                .layer(2, RnnOutputLayer(lossFunction="MCXENT", activation="SOFTMAX", nOut=self.totalUniqueCharacters))
                .backpropType(BackpropType.TruncatedBPTT).tBPTTForwardLength(50).tBPTTBackwardLength(50).build())


# ============================================================================= InceptionResNetV1
def _bn995(n, act=None):
    kw = dict(lockGammaBeta=False, decay=0.995, eps=0.001, nIn=n, nOut=n)
    if act:
        kw["activation"] = act
    return BatchNormalization(**kw)


def _irn(block, layer, i):
    return "%s-%s-%d" % (block, layer, i)


def _inception_res(g, block, scale, act_scale, inp, kind):
    """InceptionResNetHelper.inceptionV1Res{A,B,C}."""
    NW = "NO_WORKSPACE"
    same = ConvolutionMode.Same
    prev = inp
    if kind == "B":
        g.addLayer(_irn(block, "activation1", 0), _act("TANH"), inp)
        prev = _irn(block, "activation1", 0)
    for i in range(1, scale + 1):
        def cb(cn, bn, k, nin, nout, src, act=None):
            g.addLayer(_irn(block, cn, i), ConvolutionLayer(kernelSize=k, nIn=nin, nOut=nout, convolutionMode=same, cudnnAlgoMode=NW), src)
            g.addLayer(_irn(block, bn, i), _bn995(nout, act), _irn(block, cn, i))
            return _irn(block, bn, i)
        if kind == "A":
            b1 = cb("cnn1", "batch1", (1, 1), 192, 32, prev)
            b2 = cb("cnn2", "batch2", (1, 1), 192, 32, prev)
            b3 = cb("cnn3", "batch3", (3, 3), 32, 32, b2)
            b4 = cb("cnn4", "batch4", (1, 1), 192, 32, prev)
            b5 = cb("cnn5", "batch5", (3, 3), 32, 32, b4)
            b6 = cb("cnn6", "batch6", (3, 3), 32, 32, b5)
            g.addVertex(_irn(block, "merge1", i), MergeVertex(), b1, b3, b6)
            last = cb("cnn7", "batch7", (3, 3), 96, 192, _irn(block, "merge1", i))
        else:
            cin, mid, ta = (576, 128, None) if kind == "B" else (1344, 192, "TANH")
            b1 = cb("cnn1", "batch1", (1, 1), cin, mid, prev)
            b2 = cb("cnn2", "batch2", (1, 1), cin, mid, prev)
            b3 = cb("cnn3", "batch3", (1, 3), mid, mid, b2)
            b4 = cb("cnn4", "batch4", (3, 1), mid, mid, b3, ta)
            g.addVertex(_irn(block, "merge1", i), MergeVertex(), b1, b4)
            last = cb("cnn5", "batch5", (1, 1), 2 * mid, cin, _irn(block, "merge1", i), ta)
        g.addVertex(_irn(block, "scaling", i), ScaleVertex(act_scale), last)
        g.addLayer(_irn(block, "shortcut-identity", i), _act("IDENTITY"), prev)
        g.addVertex(_irn(block, "shortcut", i), ElementWiseVertex(Add), _irn(block, "scaling", i), _irn(block, "shortcut-identity", i))
        g.addLayer(block if i == scale else _irn(block, "activation", i), _act("TANH"), _irn(block, "shortcut", i))
        prev = _irn(block, "activation", i)
    return g


class InceptionResNetV1(_Zoo):
    """InceptionResNetV1.java:77-317 (FaceNet-style embedding head with center loss)."""
    DEFAULT_SHAPE = (3, 160, 160)
    embeddingSize = 128

    def _default_updater(self):
        return RmsProp(0.1, 0.96, 0.001)

    def graphBuilder(self, inp="input1"):
        am = self.cudnnAlgoMode
        same = ConvolutionMode.Same
        g = (self._base().activation("RELU").weightInit(WeightInitDistribution(TruncatedNormalDistribution(0.0, 0.5))).l2(5e-5).miniBatch(True)
             .convolutionMode(ConvolutionMode.Truncate).graphBuilder())

        def cb(cn, bn, k, s, nin, nout, src, mode=None):
            kw = {"convolutionMode": mode} if mode else {}
            g.addLayer(cn, ConvolutionLayer(kernelSize=k, stride=s, nIn=nin, nOut=nout, cudnnAlgoMode=am, **kw), src)
            g.addLayer(bn, _bn995(nout), cn)
            return bn
        cb("stem-cnn1", "stem-batch1", (3, 3), (2, 2), self.inputShape[0], 32, inp)
        cb("stem-cnn2", "stem-batch2", (3, 3), (1, 1), 32, 32, "stem-batch1")
        cb("stem-cnn3", "stem-batch3", (3, 3), (1, 1), 32, 64, "stem-batch2", same)
        g.addLayer("stem-pool4", _pool(PoolingType.MAX, 3, 2), "stem-batch3")
        cb("stem-cnn5", "stem-batch5", (1, 1), (1, 1), 64, 80, "stem-pool4")
        cb("stem-cnn6", "stem-batch6", (3, 3), (1, 1), 80, 128, "stem-batch5")
        cb("stem-cnn7", "stem-batch7", (3, 3), (2, 2), 128, 192, "stem-batch6")
        _inception_res(g, "resnetA", 5, 0.17, "stem-batch7", "A")
        cb("reduceA-cnn1", "reduceA-batch1", (3, 3), (2, 2), 192, 192, "resnetA")
        cb("reduceA-cnn2", "reduceA-batch2", (1, 1), (1, 1), 192, 128, "resnetA", same)
        cb("reduceA-cnn3", "reduceA-batch3", (3, 3), (1, 1), 128, 128, "reduceA-batch2", same)
        cb("reduceA-cnn4", "reduceA-batch4", (3, 3), (2, 2), 128, 192, "reduceA-batch3")
        g.addLayer("reduceA-pool5", _pool(PoolingType.MAX, 3, 2), "resnetA")
        g.addVertex("reduceA", MergeVertex(), "reduceA-batch1", "reduceA-batch4", "reduceA-pool5")
        _inception_res(g, "resnetB", 10, 0.10, "reduceA", "B")
        g.addLayer("reduceB-pool1", _pool(PoolingType.MAX, 3, 2), "resnetB")
        cb("reduceB-cnn2", "reduceB-batch1", (1, 1), (1, 1), 576, 256, "resnetB", same)
        cb("reduceB-cnn3", "reduceB-batch2", (3, 3), (2, 2), 256, 256, "reduceB-batch1")
        cb("reduceB-cnn4", "reduceB-batch3", (1, 1), (1, 1), 576, 256, "resnetB", same)
        cb("reduceB-cnn5", "reduceB-batch4", (3, 3), (2, 2), 256, 256, "reduceB-batch3")
        cb("reduceB-cnn6", "reduceB-batch5", (1, 1), (1, 1), 576, 256, "resnetB", same)
        cb("reduceB-cnn7", "reduceB-batch6", (3, 3), (1, 1), 256, 256, "reduceB-batch5", same)
        cb("reduceB-cnn8", "reduceB-batch7", (3, 3), (2, 2), 256, 256, "reduceB-batch6")
        g.addVertex("reduceB", MergeVertex(), "reduceB-pool1", "reduceB-batch2", "reduceB-batch4", "reduceB-batch7")
        _inception_res(g, "resnetC", 5, 0.20, "reduceB", "C")
        g.addLayer("avgpool", SubsamplingLayer(poolingType=PoolingType.AVG, kernelSize=(1, 1)), "resnetC")
        return g

    def conf(self):
        g = self.graphBuilder("input1")
        # the reference hard-codes nIn=5376 (=1344*2*2, i.e. 160x160 input); nIn is inferred here so other input sizes also work
        (g.addInputs("input1").setInputTypes(self._cnn_input())
          .addLayer("bottleneck", DenseLayer(nOut=self.embeddingSize), "avgpool")
          .addVertex("embeddings", L2NormalizeVertex([1], 1e-10), "bottleneck")
          .addLayer("outputLayer", CenterLossOutputLayer(lossFunction="NEGATIVELOGLIKELIHOOD", activation="SOFTMAX", alpha=0.9, lambda_=1e-4,
                                                         nIn=self.embeddingSize, nOut=self.numClasses), "embeddings")
          .setOutputs("outputLayer"))
        return g.build()


# ============================================================================= FaceNetNN4Small2
def _fn_bn(n):
    return BatchNormalization(lockGammaBeta=False, nIn=n, nOut=n)


def _fn_inception(g, name, nin, ks, kstride, outs, reduces, ptype, pnorm, tf, inp, poolSize=3, poolStride=1):
    """FaceNetHelper.appendGraph."""
    m = "inception-" + name
    NW = "NO_WORKSPACE"
    for i, k in enumerate(ks):
        pad = (kstride[i] // 2) * 2
        (g.addLayer("%s-cnn1-%d" % (m, i), ConvolutionLayer(kernelSize=(1, 1), stride=(1, 1), padding=(0, 0), nIn=nin, nOut=reduces[i], biasInit=0.2,
                                                             cudnnAlgoMode=NW), inp)
          .addLayer("%s-batch1-%d" % (m, i), _fn_bn(reduces[i]), "%s-cnn1-%d" % (m, i))
          .addLayer("%s-transfer1-%d" % (m, i), _act(tf), "%s-batch1-%d" % (m, i))
          .addLayer("%s-reduce1-%d" % (m, i), ConvolutionLayer(kernelSize=(k, k), stride=(kstride[i],) * 2, padding=(pad, pad), nIn=reduces[i],
                                                                nOut=outs[i], biasInit=0.2, cudnnAlgoMode=NW), "%s-transfer1-%d" % (m, i))
          .addLayer("%s-batch2-%d" % (m, i), _fn_bn(outs[i]), "%s-reduce1-%d" % (m, i))
          .addLayer("%s-transfer2-%d" % (m, i), _act(tf), "%s-batch2-%d" % (m, i)))
    merge = ["%s-transfer2-%d" % (m, i) for i in range(len(ks))]
    i = len(ks)
    if i < len(reduces):
        if ptype == PoolingType.PNORM:
            if pnorm <= 0:
                raise ValueError("p-norm must be greater than zero.")
            pl = SubsamplingLayer(poolingType=PoolingType.PNORM, kernelSize=(poolSize,) * 2, stride=(poolStride,) * 2, pnorm=pnorm)
        elif ptype == PoolingType.AVG:
            pl = SubsamplingLayer(poolingType=PoolingType.AVG, kernelSize=(poolSize,) * 2, stride=(poolStride,) * 2)
        else:
            pl = SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(poolSize,) * 2, stride=(poolStride,) * 2, padding=(1, 1))
        (g.addLayer(m + "-pool1", pl, inp)
          .addLayer(m + "-cnn2", ConvolutionLayer(kernelSize=(1, 1), stride=(1, 1), nIn=nin, nOut=reduces[i], biasInit=0.2, cudnnAlgoMode=NW), m + "-pool1")
          .addLayer(m + "-batch3", _fn_bn(reduces[i]), m + "-cnn2").addLayer(m + "-transfer3", _act(tf), m + "-batch3"))
        merge.append(m + "-transfer3")
    i += 1
    if i < len(reduces):
        (g.addLayer(m + "-reduce2", ConvolutionLayer(kernelSize=(1, 1), stride=(1, 1), nIn=nin, nOut=reduces[i], biasInit=0.2, cudnnAlgoMode=NW), inp)
          .addLayer(m + "-batch4", _fn_bn(reduces[i]), m + "-reduce2").addLayer(m + "-transfer4", _act(tf), m + "-batch4"))
        merge.append(m + "-transfer4")
    g.addVertex(m, MergeVertex(), *merge)
    return g


class FaceNetNN4Small2(_Zoo):
    """FaceNetNN4Small2.java:76-349."""
    DEFAULT_SHAPE = (3, 96, 96)
    embeddingSize = 128
    transferFunction = "RELU"

    def _default_updater(self):
        return Adam(0.1, 0.9, 0.999, 0.01)

    def conf(self):
        NW = "NO_WORKSPACE"
        tf = self.transferFunction
        g = (self._base().activation("IDENTITY").weightInit("RELU").l2(5e-5).miniBatch(True).cudnnAlgoMode(self.cudnnAlgoMode)
             .convolutionMode(ConvolutionMode.Same).graphBuilder())

        def conv(name, k, s, nin, nout, inp, pad=None):
            kw = {"padding": pad} if pad else {}
            g.addLayer(name, ConvolutionLayer(kernelSize=(k, k), stride=(s, s), nIn=nin, nOut=nout, cudnnAlgoMode=NW, **kw), inp)

        def mp(name, s, inp):
            g.addLayer(name, SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(3, 3), stride=(s, s), padding=(1, 1)), inp)
        g.addInputs("input1")
        conv("stem-cnn1", 7, 2, self.inputShape[0], 64, "input1", (3, 3))
        g.addLayer("stem-batch1", _fn_bn(64), "stem-cnn1").addLayer("stem-activation1", _act(), "stem-batch1")
        mp("stem-pool1", 2, "stem-activation1")
        g.addLayer("stem-lrn1", LocalResponseNormalization(k=1, n=5, alpha=1e-4, beta=0.75), "stem-pool1")
        conv("inception-2-cnn1", 1, 1, 64, 64, "stem-lrn1")
        g.addLayer("inception-2-batch1", _fn_bn(64), "inception-2-cnn1").addLayer("inception-2-activation1", _act(), "inception-2-batch1")
        conv("inception-2-cnn2", 3, 1, 64, 192, "inception-2-activation1", (1, 1))
        g.addLayer("inception-2-batch2", _fn_bn(192), "inception-2-cnn2").addLayer("inception-2-activation2", _act(), "inception-2-batch2")
        g.addLayer("inception-2-lrn1", LocalResponseNormalization(k=1, n=5, alpha=1e-4, beta=0.75), "inception-2-activation2")
        mp("inception-2-pool1", 2, "inception-2-lrn1")
        _fn_inception(g, "3a", 192, (3, 5), (1, 1), (128, 32), (96, 16, 32, 64), PoolingType.MAX, 0, tf, "inception-2-pool1")
        _fn_inception(g, "3b", 256, (3, 5), (1, 1), (128, 64), (96, 32, 64, 64), PoolingType.PNORM, 2, tf, "inception-3a")

        def reduce_block(p, nin, a, b, c_, d, inp, out):
            conv(p + "-1x1", 1, 1, nin, a, inp)
            g.addLayer(p + "-1x1-norm", _fn_bn(a), p + "-1x1").addLayer(p + "-transfer1", _act(tf), p + "-1x1-norm")
            conv(p + "-3x3", 3, 2, a, b, p + "-transfer1")
            g.addLayer(p + "-3x3-norm", _fn_bn(b), p + "-3x3").addLayer(p + "-transfer2", _act(tf), p + "-3x3-norm")
            conv(p + "-2-1x1", 1, 1, nin, c_, inp)
            g.addLayer(p + "-2-1x1-norm", _fn_bn(c_), p + "-2-1x1").addLayer(p + "-2-transfer3", _act(tf), p + "-2-1x1-norm")
            conv(p + "-2-5x5", 3, 2, c_, d, p + "-2-transfer3")
            g.addLayer(p + "-2-5x5-norm", _fn_bn(d), p + "-2-5x5").addLayer(p + "-2-transfer4", _act(tf), p + "-2-5x5-norm")
            mp(p + "-pool", 2, inp)
            g.addVertex(out, MergeVertex(), p + "-transfer2", p + "-2-transfer4", p + "-pool")
        reduce_block("3c", 320, 128, 256, 32, 64, "inception-3b", "inception-3c")
        _fn_inception(g, "4a", 640, (3, 5), (1, 1), (192, 64), (96, 32, 128, 256), PoolingType.PNORM, 2, tf, "inception-3c")
        reduce_block("4e", 640, 160, 256, 64, 128, "inception-4a", "inception-4e")

        def final_block(p, nin, pool_layer, inp, out, t3, last_act):
            conv(p + "-1x1", 1, 1, nin, 256, inp)
            g.addLayer(p + "-1x1-norm", _fn_bn(256), p + "-1x1").addLayer(p + "-transfer1", _act(tf), p + "-1x1-norm")
            conv(p + "-2-1x1", 1, 1, nin, 96, inp)
            g.addLayer(p + "-2-1x1-norm", _fn_bn(96), p + "-2-1x1").addLayer(p + "-2-transfer2", _act(tf), p + "-2-1x1-norm")
            conv(p + "-2-3x3", 3, 1, 96, 384, p + "-2-transfer2")
            g.addLayer(p + "-2-3x3-norm", _fn_bn(384), p + "-2-3x3").addLayer(t3, _act(tf), p + "-2-3x3-norm")
            g.addLayer(p + "-3-pool", pool_layer, inp)
            conv(p + "-3-1x1reduce", 1, 1, nin, 96, p + "-3-pool")
            g.addLayer(p + "-3-1x1reduce-norm", _fn_bn(96), p + "-3-1x1reduce").addLayer(p + "-3-transfer4", _act(last_act), p + "-3-1x1reduce-norm")
            g.addVertex(out, MergeVertex(), p + "-transfer1", t3, p + "-3-transfer4")
        final_block("5a", 1024, SubsamplingLayer(poolingType=PoolingType.PNORM, kernelSize=(3, 3), stride=(1, 1), pnorm=2), "inception-4e",
                    "inception-5a", "5a-transfer3", "RELU")
        final_block("5b", 736, SubsamplingLayer(poolingType=PoolingType.MAX, kernelSize=(3, 3), stride=(1, 1), padding=(1, 1)), "inception-5a",
                    "inception-5b", "5b-2-transfer3", tf)
        (g.addLayer("avgpool", SubsamplingLayer(poolingType=PoolingType.AVG, kernelSize=(3, 3), stride=(3, 3)), "inception-5b")
          .addLayer("bottleneck", DenseLayer(nOut=self.embeddingSize, activation="IDENTITY"), "avgpool")
          .addVertex("embeddings", L2NormalizeVertex([], 1e-6), "bottleneck")
          .addLayer("lossLayer", CenterLossOutputLayer(lossFunction="SQUARED_LOSS", activation="SOFTMAX", nOut=self.numClasses, lambda_=1e-4, alpha=0.9,
                                                       gradientNormalization=GradientNormalization.RenormalizeL2PerLayer), "embeddings")
          .setOutputs("lossLayer").setInputTypes(self._cnn_input()))
        return g.build()


# ============================================================================= NASNet
class NASNet(_Zoo):
    """NASNet-A (NASNet.java:106-188 + NASNetHelper.java).

    Deviations from the reference, which cannot build its own default configuration: ``adjustBlock`` dereferences a null
    ``inputP`` on the first cell and wires ``_avg_pool_1`` / ``_zeropad_1`` names that were never added
    (NASNetHelper.java:78,81), and ``sepConvBlock`` strides *both* separable convs so the reduction-cell branches disagree
    in size.  Here the first cell uses ``inputX`` for the missing ``inputP``, the adjust towers are wired to the layers that
    exist, and only the first separable conv of a block is strided (the published NASNet-A cell).  Layer names otherwise match."""
    numBlocks = 6
    penultimateFilters = 1056
    stemFilters = 96
    filterMultiplier = 2
    skipReduction = True

    def __init__(self, **kw):
        for k in ("numBlocks", "penultimateFilters", "stemFilters", "filterMultiplier", "skipReduction"):
            if k in kw:
                setattr(self, k, kw.pop(k))
        kw.setdefault("cacheMode", CacheMode.DEVICE)
        super().__init__(**kw)

    def _default_updater(self):
        return AdaDelta()

    @staticmethod
    def _bn():
        return BatchNormalization(eps=1e-3, gamma=0.9997)

    def _sep_block(self, g, filters, k, stride, bid, inp):
        p = "sepConvBlock" + bid
        same = ConvolutionMode.Same
        (g.addLayer(p + "_act", _act(), inp)
          .addLayer(p + "_sepconv1", SeparableConvolution2D(kernelSize=(k, k), stride=(stride, stride), nOut=filters, hasBias=False, convolutionMode=same), p + "_act")
          .addLayer(p + "_conv1_bn", self._bn(), p + "_sepconv1").addLayer(p + "_act2", _act(), p + "_conv1_bn")
          .addLayer(p + "_sepconv2", SeparableConvolution2D(kernelSize=(k, k), stride=(1, 1), nOut=filters, hasBias=False, convolutionMode=same), p + "_act2")
          .addLayer(p + "_conv2_bn", self._bn(), p + "_sepconv2"))
        c, h = self._shape[inp]
        self._shape[p + "_conv2_bn"] = (filters, -(-h // stride))
        return p + "_conv2_bn"

    def _adjust(self, g, filters, bid, inp, match):
        p = "adjustBlock" + bid
        out = inp
        if inp is None:
            inp = out = match
        same = ConvolutionMode.Same
        (ci, hi), (_, hm) = self._shape[inp], self._shape[match]
        if hm != hi:
            half = filters // 2
            (g.addLayer(p + "_relu1", _act(), inp)
              .addLayer(p + "_avgpool1", SubsamplingLayer(poolingType=PoolingType.AVG, kernelSize=(1, 1), stride=(2, 2), convolutionMode=ConvolutionMode.Truncate), p + "_relu1")
              .addLayer(p + "_conv1", _conv(1, half, 1, hasBias=False, convolutionMode=same), p + "_avgpool1")
              .addLayer(p + "_zeropad1", ZeroPaddingLayer(padding=(0, 1, 0, 1)), p + "_relu1")
              .addLayer(p + "_crop1", Cropping2D(cropping=(1, 0, 1, 0)), p + "_zeropad1")
              .addLayer(p + "_avgpool2", SubsamplingLayer(poolingType=PoolingType.AVG, kernelSize=(1, 1), stride=(2, 2), convolutionMode=ConvolutionMode.Truncate), p + "_crop1")
              .addLayer(p + "_conv2", _conv(1, filters - half, 1, hasBias=False, convolutionMode=same), p + "_avgpool2")
              .addVertex(p + "_concat1", MergeVertex(), p + "_conv1", p + "_conv2")
              .addLayer(p + "_bn1", self._bn(), p + "_concat1"))
            out = p + "_bn1"
            ci, hi = filters, (hi - 1) // 2 + 1
            self._shape[out] = (ci, hi)
        if ci != filters:
            (g.addLayer(p + "_projection_relu", _act(), out)
              .addLayer(p + "_projection_conv", _conv(1, filters, 1, hasBias=False, convolutionMode=same), p + "_projection_relu")
              .addLayer(p + "_projection_bn", self._bn(), p + "_projection_conv"))
            out = p + "_projection_bn"
            self._shape[out] = (filters, hi)
        return out

    def _top(self, g, p, filters, inputX):
        (g.addLayer(p + "_relu1", _act(), inputX)
          .addLayer(p + "_conv1", _conv(1, filters, 1, hasBias=False, convolutionMode=ConvolutionMode.Same), p + "_relu1")
          .addLayer(p + "_bn1", self._bn(), p + "_conv1"))
        self._shape[p + "_bn1"] = (filters, self._shape[inputX][1])

    def _normal(self, g, filters, bid, inputX, inputP):
        p = "normalA" + bid
        adj = self._adjust(g, filters, p, inputP, inputX)
        self._top(g, p, filters, inputX)
        h, same = p + "_bn1", ConvolutionMode.Same
        ap = lambda: SubsamplingLayer(poolingType=PoolingType.AVG, kernelSize=(3, 3), stride=(1, 1), convolutionMode=same)
        g.addVertex(p + "_add1", ElementWiseVertex(Add), self._sep_block(g, filters, 5, 1, p + "_left1", h), self._sep_block(g, filters, 3, 1, p + "_right1", adj))
        g.addVertex(p + "_add2", ElementWiseVertex(Add), self._sep_block(g, filters, 5, 1, p + "_left2", adj), self._sep_block(g, filters, 3, 1, p + "_right2", adj))
        g.addLayer(p + "_left3", ap(), h).addVertex(p + "_add3", ElementWiseVertex(Add), p + "_left3", adj)
        g.addLayer(p + "_left4", ap(), adj).addLayer(p + "_right4", ap(), adj).addVertex(p + "_add4", ElementWiseVertex(Add), p + "_left4", p + "_right4")
        g.addVertex(p + "_add5", ElementWiseVertex(Add), self._sep_block(g, filters, 3, 1, p + "_left5", h), h)
        g.addVertex(p, MergeVertex(), adj, p + "_add1", p + "_add2", p + "_add3", p + "_add4", p + "_add5")
        self._shape[p] = (6 * filters, self._shape[h][1])
        return p, inputX

    def _reduction(self, g, filters, bid, inputX, inputP):
        p = "reductionA" + bid
        adj = self._adjust(g, filters, p, inputP, inputX)
        self._top(g, p, filters, inputX)
        h, same = p + "_bn1", ConvolutionMode.Same
        pool = lambda kind, s: SubsamplingLayer(poolingType=kind, kernelSize=(3, 3), stride=(s, s), convolutionMode=same)
        g.addVertex(p + "_add1", ElementWiseVertex(Add), self._sep_block(g, filters, 5, 2, p + "_left1", h), self._sep_block(g, filters, 7, 2, p + "_right1", adj))
        g.addLayer(p + "_left2", pool(PoolingType.MAX, 2), h)
        g.addVertex(p + "_add2", ElementWiseVertex(Add), p + "_left2", self._sep_block(g, filters, 7, 2, p + "_right2", adj))
        g.addLayer(p + "_left3", pool(PoolingType.AVG, 2), h)
        g.addVertex(p + "_add3", ElementWiseVertex(Add), p + "_left3", self._sep_block(g, filters, 5, 2, p + "_right3", adj))
        hh = -(-self._shape[h][1] // 2)
        self._shape[p + "_add1"] = (filters, hh)
        g.addLayer(p + "_left4", pool(PoolingType.AVG, 1), p + "_add1").addVertex(p + "_add4", ElementWiseVertex(Add), p + "_add2", p + "_left4")
        left5 = self._sep_block(g, filters, 3, 1, p + "_left5", p + "_add1")
        g.addLayer(p + "_right5", pool(PoolingType.MAX, 2), h).addVertex(p + "_add5", ElementWiseVertex(Add), left5, p + "_right5")
        g.addVertex(p, MergeVertex(), p + "_add2", p + "_add3", p + "_add4", p + "_add5")
        self._shape[p] = (4 * filters, hh)
        return p, inputX

    def graphBuilder(self):
        if self.penultimateFilters % 24 != 0:
            raise ValueError("For NASNet-A models penultimate filters must be divisible by 24. Current value is %d" % self.penultimateFilters)
        filters, fm, nb = self.penultimateFilters // 24, self.filterMultiplier, self.numBlocks
        g = (self._base().weightInit(self.weightInit or "RELU").l2(5e-5).miniBatch(True).cudnnAlgoMode(self.cudnnAlgoMode)
             .convolutionMode(ConvolutionMode.Truncate).graphBuilder())
        s = 1 if self.skipReduction else 2
        g.addLayer("stem_conv1", _conv(3, self.stemFilters, s, hasBias=False, cudnnAlgoMode=self.cudnnAlgoMode), "input")
        g.addLayer("stem_bn1", self._bn(), "stem_conv1")
        hin = (self.inputShape[1] - 3) // s + 1
        self._shape = {"stem_bn1": (self.stemFilters, hin), "stem_conv1": (self.stemFilters, hin)}
        x, pz = "stem_bn1", None
        if not self.skipReduction:
            x, pz = self._reduction(g, int(self.stemFilters // fm ** 2), "stem1", "stem_bn1", None)
            x, pz = self._reduction(g, int(self.stemFilters // fm), "stem2", x, pz)
        for i in range(nb):
            x, pz = self._normal(g, filters, str(i), x, pz)
        x, p0 = self._reduction(g, filters * fm, "reduce%d" % nb, x, pz)
        pz = p0 if not self.skipReduction else pz
        for i in range(nb):
            x, pz = self._normal(g, filters * fm, str(i + nb + 1), x, pz)
        x, p0 = self._reduction(g, filters * fm ** 2, "reduce%d" % (2 * nb), x, pz)
        pz = p0 if not self.skipReduction else pz
        for i in range(nb):
            x, pz = self._normal(g, filters * fm ** 2, str(i + 2 * nb + 1), x, pz)
        (g.addLayer("act", _act(), x).addLayer("avg_pool", GlobalPoolingLayer(poolingType=PoolingType.AVG), "act")
          .addLayer("output", OutputLayer(lossFunction="MCXENT", activation="SOFTMAX", nOut=self.numClasses), "avg_pool").setOutputs("output"))
        return g

    def conf(self):
        return self.graphBuilder().addInputs("input").setInputTypes(self._cnn_input()).build()


ALL_MODELS = {c.__name__: c for c in (AlexNet, SimpleCNN, VGG16, VGG19, Darknet19, TinyYOLO, YOLO2, SqueezeNet, UNet, Xception, TextGenerationLSTM,
                                      InceptionResNetV1, FaceNetNN4Small2, NASNet)}
