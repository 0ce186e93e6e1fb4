"""CPU tests of the network engine: LeNet/MLP training, config JSON round-trip, fused vs unfused
ComputationGraph, gradient checks (fp64, central differences — mirrors DL4J GradientCheckUtil:
eps 1e-6, maxRelError 1e-3), ModelSerializer round trip."""
import io
import os

import pytest
import torch

from deeplearning4j_amd.nd4j.learning import Adam, Sgd, Nesterovs, RmsProp, NoOp
from deeplearning4j_amd.nn.conf.base import InputType, ConvolutionMode
from deeplearning4j_amd.nn.conf.configuration import (NeuralNetConfiguration, MultiLayerConfiguration,
                                                      ComputationGraphConfiguration)
from deeplearning4j_amd.nn.conf.graph import ElementWiseVertex, MergeVertex
from deeplearning4j_amd.nn.conf.layers import (DenseLayer, OutputLayer, ConvolutionLayer, SubsamplingLayer,
                                               BatchNormalization, ActivationLayer, LSTM, GravesLSTM, RnnOutputLayer,
                                               GlobalPoolingLayer, EmbeddingLayer, DropoutLayer, ZeroPaddingLayer,
                                               SimpleRnn, Bidirectional, LastTimeStep)
from deeplearning4j_amd.nn.graph.computation_graph import ComputationGraph
from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
from deeplearning4j_amd.utils.gradient_check import GradientCheckUtil


def _onehot(n, c, seed=0):
    g = torch.Generator().manual_seed(seed)
    y = torch.zeros(n, c, dtype=torch.float64)
    y[torch.arange(n), torch.randint(0, c, (n,), generator=g)] = 1
    return y


def test_mlp_learns():
    conf = (NeuralNetConfiguration.Builder().seed(42).updater(Adam(0.05)).weightInit("XAVIER").list()
            .layer(DenseLayer(nIn=4, nOut=16, activation="TANH"))
            .layer(OutputLayer(nIn=16, nOut=3, activation="SOFTMAX", lossFunction="MCXENT")).build())
    net = MultiLayerNetwork(conf, device="cpu").init()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(60, 4, generator=g)
    lab = (x[:, 0] > 0).long() + (x[:, 1] > 0).long()
    y = torch.zeros(60, 3)
    y[torch.arange(60), lab] = 1
    net.fit(x, y)
    s0 = net.score()
    for _ in range(150):
        net.fit(x, y)
    assert net.score() < 0.5 * s0
    acc = (net.output(x).argmax(1) == lab).float().mean()
    assert acc > 0.9


def test_json_roundtrip_mln_and_cg():
    conf = (NeuralNetConfiguration.Builder().seed(7).updater(Nesterovs(0.01, 0.9)).l2(1e-4).list()
            .layer(ConvolutionLayer(kernelSize=(3, 3), nOut=4, activation="RELU"))
            .layer(SubsamplingLayer(kernelSize=(2, 2), stride=(2, 2)))
            .layer(BatchNormalization())
            .layer(DenseLayer(nOut=8, activation="TANH"))
            .layer(OutputLayer(nOut=3, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.convolutionalFlat(8, 8, 1)).build())
    js = conf.toJson()
    conf2 = MultiLayerConfiguration.fromJson(js)
    assert conf2.toJson() == js
    assert "org.deeplearning4j.nn.conf.layers.ConvolutionLayer" in js
    n1 = MultiLayerNetwork(conf, device="cpu").init()
    n2 = MultiLayerNetwork(conf2, device="cpu").init()
    assert torch.equal(n1.params(), n2.params())
    yml = conf.toYaml()
    assert MultiLayerConfiguration.fromYaml(yml).toJson() == js

    g = (NeuralNetConfiguration.Builder().seed(3).updater(Sgd(0.1)).graphBuilder().addInputs("a", "b")
         .addLayer("d1", DenseLayer(nIn=3, nOut=5, activation="RELU"), "a")
         .addLayer("d2", DenseLayer(nIn=4, nOut=5, activation="RELU"), "b")
         .addVertex("add", ElementWiseVertex("Add"), "d1", "d2")
         .addVertex("m", MergeVertex(), "add", "d1")
         .addLayer("out", OutputLayer(nIn=10, nOut=2, activation="SOFTMAX", lossFunction="MCXENT"), "m")
         .setOutputs("out").build())
    js = g.toJson()
    g2 = ComputationGraphConfiguration.fromJson(js)
    assert g2.toJson() == js
    cg = ComputationGraph(g2, device="cpu").init()
    out = cg.output(torch.rand(5, 3), torch.rand(5, 4))[0]
    assert out.shape == (5, 2)
    assert torch.allclose(out.sum(1), torch.ones(5), atol=1e-5)


def _gc(net, x, y, **kw):
    assert GradientCheckUtil.checkGradients(net, x, y, **kw)


def test_gradcheck_dense_conv_bn():
    conf = (NeuralNetConfiguration.Builder().seed(12345).dataType("DOUBLE").updater(NoOp()).weightInit("XAVIER")
            .l2(0.01).l1(0.005).list()
            .layer(ConvolutionLayer(kernelSize=(3, 3), nOut=3, activation="TANH", convolutionMode=ConvolutionMode.Same))
            .layer(BatchNormalization())
            .layer(ActivationLayer(activation="RELU"))
            .layer(SubsamplingLayer(poolingType="MAX", kernelSize=(2, 2), stride=(2, 2)))
            .layer(ConvolutionLayer(kernelSize=(2, 2), nOut=2, activation="IDENTITY"))
            .layer(BatchNormalization(useLogStd=False))
            .layer(SubsamplingLayer(poolingType="AVG", kernelSize=(2, 2), stride=(1, 1)))
            .layer(DenseLayer(nOut=6, activation="SIGMOID"))
            .layer(OutputLayer(nOut=3, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.convolutional(8, 8, 2)).build())
    net = MultiLayerNetwork(conf, device="cpu").init()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 2, 8, 8, generator=g, dtype=torch.float64)
    _gc(net, x, _onehot(6, 3), exclude=("mean", "var", "log10stdev"))


@pytest.mark.parametrize("layer", ["lstm", "graves", "simple", "bidir"])
def test_gradcheck_recurrent(layer):
    n = 4
    rl = {"lstm": LSTM(nOut=n, activation="TANH"), "graves": GravesLSTM(nOut=n, activation="TANH"),
          "simple": SimpleRnn(nOut=n, activation="TANH"),
          "bidir": Bidirectional("CONCAT", LSTM(nOut=n, activation="TANH"))}[layer]
    conf = (NeuralNetConfiguration.Builder().seed(12345).dataType("DOUBLE").updater(NoOp()).weightInit("XAVIER").list()
            .layer(rl)
            .layer(RnnOutputLayer(nOut=3, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.recurrent(3, 5)).build())
    net = MultiLayerNetwork(conf, device="cpu").init()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 5, generator=g, dtype=torch.float64)
    y = torch.zeros(2, 3, 5, dtype=torch.float64)
    idx = torch.randint(0, 3, (2, 5), generator=g)
    for b in range(2):
        for t in range(5):
            y[b, idx[b, t], t] = 1
    _gc(net, x, y)


def test_gradcheck_lstm_masked_lasttimestep():
    conf = (NeuralNetConfiguration.Builder().seed(1).dataType("DOUBLE").updater(NoOp()).list()
            .layer(LastTimeStep(GravesLSTM(nOut=4, activation="TANH")))
            .layer(OutputLayer(nOut=2, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.recurrent(3, 6)).build())
    net = MultiLayerNetwork(conf, device="cpu").init()
    x = torch.randn(3, 3, 6, dtype=torch.float64)
    mask = torch.tensor([[1, 1, 1, 1, 1, 1], [1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 0]], dtype=torch.float64)
    assert GradientCheckUtil.checkGradients(net, x, _onehot(3, 2), featuresMask=mask)


def test_cg_fused_equals_unfused():
    def build(fuse):
        g = (NeuralNetConfiguration.Builder().seed(9).dataType("DOUBLE").updater(Sgd(0.1)).activation("IDENTITY")
             .graphBuilder().addInputs("in").setInputTypes(InputType.convolutional(8, 8, 3))
             .addLayer("c1", ConvolutionLayer(kernelSize=(3, 3), nOut=4, convolutionMode="Same"), "in")
             .addLayer("b1", BatchNormalization(), "c1")
             .addLayer("a1", ActivationLayer(activation="RELU"), "b1")
             .addLayer("c2", ConvolutionLayer(kernelSize=(1, 1), nOut=4), "a1")
             .addLayer("b2", BatchNormalization(), "c2")
             .addLayer("cs", ConvolutionLayer(kernelSize=(1, 1), nOut=4), "in")
             .addLayer("bs", BatchNormalization(), "cs")
             .addVertex("add", ElementWiseVertex("Add"), "b2", "bs")
             .addLayer("r", ActivationLayer(activation="RELU"), "add")
             .addLayer("c3", ConvolutionLayer(kernelSize=(1, 1), nOut=4), "r")
             .addLayer("b3", BatchNormalization(), "c3")
             .addVertex("add2", ElementWiseVertex("Add"), "b3", "r")
             .addLayer("r2", ActivationLayer(activation="RELU"), "add2")
             .addLayer("gp", GlobalPoolingLayer(poolingType="AVG"), "r2")
             .addLayer("out", OutputLayer(nOut=3, activation="SOFTMAX", lossFunction="MCXENT"), "gp")
             .setOutputs("out").build())
        cg = ComputationGraph(g, device="cpu")
        cg.fusion_enabled = fuse
        return cg.init()
    a, b = build(True), build(False)
    assert len(a._plan) < len(b._plan)
    x = torch.randn(5, 3, 8, 8, dtype=torch.float64)
    y = _onehot(5, 3)
    a.computeGradientAndScore([x], [y])
    b.computeGradientAndScore([x], [y])
    assert abs(float(a._score) - float(b._score)) < 1e-10
    assert torch.allclose(a._grads, b._grads, atol=1e-9)
    for _ in range(3):
        a.fit(x, y)
        b.fit(x, y)
    assert torch.allclose(a.params(), b.params(), atol=1e-9)
    assert torch.allclose(a.output(x)[0], b.output(x)[0], atol=1e-9)
    assert GradientCheckUtil.checkGradients(a, [x], [y], exclude=("mean", "var", "log10stdev", "b"))


def test_model_serializer_roundtrip(tmp_path):
    conf = (NeuralNetConfiguration.Builder().seed(11).updater(Adam(0.01)).list()
            .layer(DenseLayer(nIn=5, nOut=7, activation="RELU"))
            .layer(OutputLayer(nIn=7, nOut=2, activation="SOFTMAX", lossFunction="MCXENT")).build())
    net = MultiLayerNetwork(conf, device="cpu").init()
    x, y = torch.rand(8, 5), _onehot(8, 2).float()
    for _ in range(3):
        net.fit(x, y)
    p = tmp_path / "m.zip"
    net.save(str(p), True)
    import zipfile
    names = set(zipfile.ZipFile(str(p)).namelist())
    assert {"configuration.json", "coefficients.bin", "updaterState.bin"} <= names
    net2 = MultiLayerNetwork.load(str(p), True, device="cpu")
    assert torch.equal(net.params(), net2.params())
    assert torch.equal(net.getUpdaterState(), net2.getUpdaterState())
    assert net2.getIterationCount() == 3
    net.fit(x, y)
    net2.fit(x, y)
    assert torch.allclose(net.params(), net2.params(), atol=1e-7)


def test_samediff_output_layer_vertex_and_lambda_vertex_train():
    """User-defined SameDiffOutputLayer / SameDiffVertex / SameDiffLambdaVertex (reference: TestSameDiffOutput, TestSameDiffVertex,
    TestSameDiffLambda in deeplearning4j-core samediff tests): parameters are allocated in the flat vector, take the global
    updater and train."""
    from deeplearning4j_amd.nn.conf.graph import SameDiffVertex, SameDiffLambdaVertex
    from deeplearning4j_amd.nn.conf.layers import SameDiffOutputLayer

    class MseOut(SameDiffOutputLayer):
        def defineParameters(self):
            return {"W": (self.nIn, self.nOut), "b": (1, self.nOut)}

        def defineLayer(self, x, labels, params):
            y = torch.tanh(x @ params["W"] + params["b"])
            loss = None if labels is None else ((y - labels) ** 2).sum(1)
            return y, loss

    class ScaleAddVertex(SameDiffVertex):
        def __init__(self, n):
            self.n = n

        def defineParameters(self):
            return {"W": (self.n, self.n)}

        def defineVertex(self, inputs, params):
            return inputs[0] @ params["W"] + inputs[1]

    torch.manual_seed(0)
    conf = (NeuralNetConfiguration.Builder().seed(7).updater(Adam(0.02)).weightInit("XAVIER").graphBuilder()
            .addInputs("in")
            .addLayer("d0", DenseLayer(nIn=4, nOut=6, activation="TANH"), "in")
            .addLayer("d1", DenseLayer(nIn=4, nOut=6, activation="TANH"), "in")
            .addVertex("sd", ScaleAddVertex(6), "d0", "d1")
            .addVertex("lam", SameDiffLambdaVertex(lambda a, b: a * 0.5 + b), "sd", "d0")
            .addLayer("out", MseOut(nIn=6, nOut=2), "lam")
            .setOutputs("out").build())
    net = ComputationGraph(conf, device="cpu").init()
    assert net.numParams() == 2 * (4 * 6 + 6) + 36 + (6 * 2 + 2)
    x = torch.randn(32, 4)
    y = torch.tanh(x[:, :2] * 0.7 - x[:, 2:] * 0.3)
    net.fit([x], [y])
    s0 = net.score()
    w_vertex0 = net._impls["sd"].params["W"].clone()
    for _ in range(150):
        net.fit([x], [y])
    assert net.score() < 0.25 * s0
    assert not torch.equal(w_vertex0, net._impls["sd"].params["W"])          # vertex parameters are updated too
    out = net.outputSingle(x)
    assert out.shape == (32, 2) and float(((out - y) ** 2).sum(1).mean()) < 0.25 * s0 + 1e-3
    # numerical gradient check through the autograd-backed pieces
    conf2 = (NeuralNetConfiguration.Builder().seed(3).updater(NoOp()).dataType("DOUBLE").graphBuilder()
             .addInputs("in").addLayer("d0", DenseLayer(nIn=3, nOut=4, activation="TANH"), "in")
             .addVertex("sd", ScaleAddVertex(4), "d0", "d0").addLayer("out", MseOut(nIn=4, nOut=2), "sd").setOutputs("out").build())
    n2 = ComputationGraph(conf2, device="cpu").init()
    assert GradientCheckUtil.checkGradients(n2, [torch.randn(5, 3, dtype=torch.float64)], [torch.rand(5, 2, dtype=torch.float64)], epsilon=1e-6, maxRelError=1e-5)


def test_grad_norm_table_equals_layerwise_gradient_normalization():
    """The range-table form of GradientNormalization (one launch pair on the GPU) is the per-layer / per-parameter reference math."""
    from deeplearning4j_amd.nn.conf.base import GradientNormalization as GN
    from deeplearning4j_amd.nd4j.learning import Sgd
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import DenseLayer, OutputLayer
    from deeplearning4j_amd.nn.conf.base import InputType
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    for mode in (GN.RenormalizeL2PerLayer, GN.RenormalizeL2PerParamType, GN.ClipElementWiseAbsoluteValue, GN.ClipL2PerLayer,
                 GN.ClipL2PerParamType):
        conf = (NeuralNetConfiguration.Builder().seed(3).updater(Sgd(0.1)).gradientNormalization(mode)
                .gradientNormalizationThreshold(0.01).list()
                .layer(DenseLayer(nIn=6, nOut=9, activation="TANH")).layer(OutputLayer(nIn=9, nOut=3, activation="SOFTMAX", lossFunction="MCXENT"))
                .setInputType(InputType.feedForward(6)).build())
        net = MultiLayerNetwork(conf, device="cpu").init()
        assert net._gn_table is not None and len(net._gn_table.ranges) == (2 if "Layer" in mode else 4)
        x = torch.randn(10, 6, generator=torch.Generator().manual_seed(1))
        y = torch.eye(3)[torch.arange(10) % 3]
        net.computeGradientAndScore(x, y)
        g0 = net._grads.clone()
        net._pre_apply_gradient_normalization()
        got = net._grads.clone()
        net._grads.copy_(g0)
        tbl, net._gn_table = net._gn_table, None           # the layer-by-layer reference implementation
        net._pre_apply_gradient_normalization()
        net._gn_table = tbl
        assert torch.allclose(got, net._grads, rtol=1e-6, atol=1e-9), mode
        assert not torch.equal(got, g0)


def test_learning_rate_schedule_is_followed_on_cpu():
    from deeplearning4j_amd.nd4j.learning import Sgd, StepSchedule
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import OutputLayer
    from deeplearning4j_amd.nn.conf.base import InputType
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    conf = (NeuralNetConfiguration.Builder().seed(3).updater(Sgd(learningRateSchedule=StepSchedule("ITERATION", 0.1, 0.5, 2))).list()
            .layer(OutputLayer(nIn=4, nOut=2, activation="SOFTMAX", lossFunction="MCXENT")).setInputType(InputType.feedForward(4)).build())
    net = MultiLayerNetwork(conf, device="cpu").init()
    x = torch.randn(6, 4, generator=torch.Generator().manual_seed(1))
    y = torch.eye(2)[torch.arange(6) % 2]
    seen = []
    for _ in range(5):
        p0 = net.params().clone()
        net.fit(x, y)
        g = net._grads.clone() / 6
        step = (p0.view(-1) - net.params().view(-1))
        seen.append(float(step.norm() / g.norm()))
    assert [round(v, 6) for v in seen] == [0.1, 0.1, 0.05, 0.05, 0.025]


def _lstm_chain_graph(device, dtype=None, n=8):
    from deeplearning4j_amd.nn.conf.layers import RnnOutputLayer
    b = NeuralNetConfiguration.Builder().seed(11).updater(Sgd(0.1))
    if dtype:
        b = b.computeDataType(dtype)
    g = (b.graphBuilder().addInputs("in").setInputTypes(InputType.recurrent(6, 12))
         .addLayer("l0", LSTM(nIn=6, nOut=n, activation="TANH"), "in")
         .addLayer("l1", LSTM(nIn=n, nOut=n, activation="TANH"), "l0")
         .addLayer("l2", LSTM(nIn=n, nOut=n, activation="TANH"), "l1")
         .addLayer("side", LSTM(nIn=n, nOut=n, activation="TANH"), "l2")
         .addLayer("out", RnnOutputLayer(nIn=n, nOut=3, activation="SOFTMAX", lossFunction="MCXENT"), "side")
         .setOutputs("out").build())
    return ComputationGraph(g, device=device).init()


def test_graph_merges_lstm_chain_into_one_plan_node():
    """A chain of LSTM vertices becomes one plan node (nn/graph/computation_graph.py _LstmStackNode; pipelined over HIP streams on the GPU,
    vertex by vertex here): same score and gradients as the unfused plan, and feedForward still returns every vertex."""
    from deeplearning4j_amd.nn.graph.computation_graph import _LstmStackNode
    cg = _lstm_chain_graph("cpu")
    stacks = [nd for nd in cg._plan if isinstance(nd, _LstmStackNode)]
    assert len(stacks) == 1 and stacks[0].names == ["l0", "l1", "l2", "side"] and len(cg._plan) == 2
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, 12, generator=g)
    y = torch.nn.functional.one_hot(torch.randint(0, 3, (4, 12), generator=g), 3).permute(0, 2, 1).float()
    cg.computeGradientAndScore([x], [y])
    s1, g1 = float(cg._score), cg._grads.clone()
    ref = _lstm_chain_graph("cpu")
    ref.fusion_enabled = False
    ref._plan = ref._build_plan()
    assert not any(isinstance(nd, _LstmStackNode) for nd in ref._plan)
    ref.computeGradientAndScore([x], [y])
    assert abs(s1 - float(ref._score)) < 1e-6 and torch.allclose(g1, ref._grads, atol=1e-6)
    acts = cg.feedForward(x)
    assert all(k in acts for k in ("l0", "l1", "l2", "side", "out"))
