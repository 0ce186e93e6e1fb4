"""Realistic training configurations keep the captured (HIP-graph) training step: learning-rate schedules through the device-resident
schedule table, gradient normalization through the two-launch range kernels, bf16 / fp16 parameter vectors through the typed updater
kernel. Each case: captured run == eager run on the GPU, and both follow the CPU network (PyTorch math)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cnn(dev, updater, gn=None, thr=1.0, dataType=None, capture=False):
    from deeplearning4j_amd.nn.conf.base import InputType, ConvolutionMode
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import ConvolutionLayer, BatchNormalization, DenseLayer, OutputLayer, SubsamplingLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    b = NeuralNetConfiguration.Builder().seed(11).updater(updater).weightInit("XAVIER").convolutionMode(ConvolutionMode.Same)
    if gn:
        b = b.gradientNormalization(gn).gradientNormalizationThreshold(thr)
    if dataType:
        b = b.dataType(dataType)
    conf = (b.list()
            .layer(ConvolutionLayer(kernelSize=(3, 3), nIn=8, nOut=16, activation="IDENTITY"))
            .layer(BatchNormalization())
            .layer(SubsamplingLayer(kernelSize=(2, 2), stride=(2, 2)))
            .layer(DenseLayer(nOut=64, activation="TANH"))
            .layer(OutputLayer(nOut=8, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.convolutional(8, 8, 8)).build())
    net = MultiLayerNetwork(conf, device=dev).init()
    if capture:
        net.enableGraphCapture(True)
    return net


def _data(n=32):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, 8, 8, 8, generator=g)
    y = torch.eye(8)[torch.randint(0, 8, (n,), generator=g)]
    return x, y


@pytest.mark.parametrize("gn", [None, "ClipL2PerLayer", "RenormalizeL2PerParamType", "ClipElementWiseAbsoluteValue"])
def test_schedule_and_gradient_normalization_stay_captured(gn):
    from deeplearning4j_amd.nd4j.learning import Nesterovs, StepSchedule
    mk = lambda: Nesterovs(0.05, 0.9, learningRateSchedule=StepSchedule("ITERATION", 0.05, 0.5, 4))
    thr = 0.05 if gn and "Clip" in gn else 1.0
    cpu = _cnn("cpu", mk(), gn, thr)
    eager = _cnn("cuda", mk(), gn, thr)
    cap = _cnn("cuda", mk(), gn, thr, capture=True)
    x, y = _data()
    for it in range(14):
        cpu.fit(x, y)
        eager.fit(x.cuda(), y.cuda())
        cap.fit(x.cuda(), y.cuda())
    assert cap._graph_step.graph_a is not None, "a learning-rate schedule / gradient normalization forced the eager path"
    pe, pc, pr = eager.params().cpu().view(-1), cap.params().cpu().view(-1), cpu.params().view(-1)
    moved = float((pr - _cnn("cpu", mk(), gn, thr).params().view(-1)).norm())
    assert moved > 1e-3
    assert float((pe - pc).norm()) < 1e-4 * moved, float((pe - pc).norm())
    assert float((pe - pr).norm()) < 2e-3 * moved, float((pe - pr).norm())
    # the schedule really was applied: 14 iterations with a step of 4 -> rate 0.05 * 0.5**3 for the last ones
    assert abs(cap.getLearningRate(0) - 0.05 * 0.5 ** 3) < 1e-9 or abs(eager._updater.segments[0].lr - 0.05 * 0.5 ** 3) < 1e-9


@pytest.mark.parametrize("dt", ["HALF"])      # the legal reduced-precision network dtype (NeuralNetConfiguration.java:1149-1154)
def test_reduced_precision_parameter_vectors_update_in_the_hip_kernel(dt, monkeypatch):
    from deeplearning4j_amd.nd4j.learning import Nesterovs
    from deeplearning4j_amd.ops import updater as U
    # momentum SGD: Adam's second-moment state (g^2 ~ 1e-7) is below fp16 resolution without fp32 masters, on any framework
    net = _cnn("cuda", Nesterovs(0.05, 0.9), "ClipL2PerLayer", 0.5, dataType=dt)
    assert net.params().dtype == torch.float16

    def no_reference(*a, **k):
        raise AssertionError("per-segment PyTorch reference updater ran on the GPU")
    monkeypatch.setattr(U, "apply_segment_reference", no_reference)
    x, y = _data()
    s0 = None
    for _ in range(25):
        net.fit(x.cuda(), y.cuda())
        s0 = s0 if s0 is not None else net.score()
    assert net.score() < s0 - 0.03, (s0, net.score())          # 16-bit weights move in coarse steps; the point is that they move, in the kernel


def test_grad_norm_table_matches_per_layer_reference_on_gpu():
    from deeplearning4j_amd.ops.updater import GradNormTable
    g = torch.Generator().manual_seed(3)
    flat = torch.randn(50000, generator=g) * 3
    ranges = [(0, 7000, 1, 1.0), (7000, 123, 3, 0.5), (7123, 20000, 3, 1e6), (27123, 5, 2, 0.25), (27128, 22872, 1, 1.0)]
    ref = flat.clone()
    GradNormTable(ranges, "cpu").run(ref)
    got = flat.cuda()
    t = GradNormTable(ranges, "cuda")
    for _ in range(2):                                   # second run on fresh data: the accumulator is re-zeroed per call
        cur = flat.cuda()
        t.run(cur)
    assert torch.allclose(cur.cpu(), ref, rtol=1e-5, atol=1e-6)
