"""GPU numerics of the recurrent paths: native LSTM sequence loop (LSTM / GravesLSTM with peepholes, masks, reverse direction) and the
fused GRU cell / native gru loop, each against plain PyTorch fp32 math of the same equations (the CPU implementation of the layer)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(kind, nIn, n, dtype, dev, seed=3):
    from deeplearning4j_amd.nd4j.learning import Sgd
    from deeplearning4j_amd.nn.conf.base import InputType
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import LSTM, GravesLSTM, GravesBidirectionalLSTM, RnnOutputLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    b = NeuralNetConfiguration.Builder().seed(seed).updater(Sgd(0.1)).weightInit("XAVIER")
    if dtype:
        b = b.computeDataType(dtype)
    L = {"lstm": LSTM, "graves": GravesLSTM, "bidir": GravesBidirectionalLSTM}[kind]
    conf = (b.list().layer(L(nIn=nIn, nOut=n, activation="TANH")).layer(L(nIn=n, nOut=n, activation="TANH"))
            .layer(RnnOutputLayer(nIn=n, nOut=5, activation="SOFTMAX", lossFunction="MCXENT"))
            .setInputType(InputType.recurrent(nIn, 9)).build())
    return MultiLayerNetwork(conf, device=dev).init()


@pytest.mark.parametrize("kind", ["lstm", "graves", "bidir"])
@pytest.mark.parametrize("masked", [False, True])
def test_lstm_native_sequence_matches_cpu_fp32(kind, masked):
    a = _net(kind, 12, 16, None, "cpu")
    b = _net(kind, 12, 16, None, "cuda")
    assert torch.equal(a.params(), b.params().cpu())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 12, 9, generator=g)
    y = torch.nn.functional.one_hot(torch.randint(0, 5, (6, 9), generator=g), 5).permute(0, 2, 1).float()
    fm = lm = None
    if masked:
        lens = torch.tensor([9, 7, 9, 3, 5, 8])
        fm = (torch.arange(9).unsqueeze(0) < lens.unsqueeze(1)).float()
        lm = fm.clone()
    a.computeGradientAndScore(x, y, fm, lm)
    b.computeGradientAndScore(x.cuda(), y.cuda(), None if fm is None else fm.cuda(), None if lm is None else lm.cuda())
    assert abs(float(a._score) - float(b._score)) < 1e-4 * max(1.0, abs(float(a._score)))
    ga, gb = a._grads, b._grads.cpu()
    assert (ga - gb).abs().max().item() < 1e-4 * max(1.0, ga.abs().max().item()), (ga - gb).abs().max().item()


def test_lstm_native_sequence_is_used_and_bf16_close(monkeypatch):
    from deeplearning4j_amd.ops import lstm as R
    calls = {"n": 0}
    real = R.seq_forward

    def spy(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(R, "seq_forward", spy)
    a = _net("lstm", 16, 32, None, "cpu")
    b = _net("lstm", 16, 32, "BFLOAT16", "cuda")
    g = torch.Generator().manual_seed(2)
    x = torch.randn(8, 16, 9, generator=g)
    y = torch.nn.functional.one_hot(torch.randint(0, 5, (8, 9), generator=g), 5).permute(0, 2, 1).float()
    a.computeGradientAndScore(x, y)
    b.computeGradientAndScore(x.cuda(), y.cuda())
    assert calls["n"] == 2
    ga, gb = a._grads.double(), b._grads.cpu().double()
    cos = float((ga * gb).sum() / (ga.norm() * gb.norm()))
    assert cos > 0.995, cos


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gru_op_native_loop_matches_reference_math(dtype):
    from deeplearning4j_amd.nd4j.ops.custom import exec_op
    from deeplearning4j_amd.nd4j.ops import impl_nn
    T, B, nIn, n = 7, 5, 16, 24
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randn(T, B, nIn, device="cuda", generator=g).to(dtype)
    h0 = torch.randn(B, n, device="cuda", generator=g).to(dtype)
    Wx = (torch.randn(nIn, 3 * n, device="cuda", generator=g) / 4).to(dtype)
    Wh = (torch.randn(n, 3 * n, device="cuda", generator=g) / 5).to(dtype)
    b = torch.randn(3 * n, device="cuda", generator=g).to(dtype)
    ref_in = [v.float().detach().clone().requires_grad_(True) for v in (x, h0, Wx, Wh, b)]
    ref = impl_nn._gru(ref_in, [], [], [], [])
    dH = torch.randn_like(ref)
    rg = torch.autograd.grad(ref, ref_in, dH)
    ins = [v.detach().clone().requires_grad_(True) for v in (x, h0, Wx, Wh, b)]
    out = exec_op("gru", ins)[0]
    assert out.dtype == dtype and type(out.grad_fn).__name__.startswith("_GruSeq")
    tol = 1e-4 if dtype == torch.float32 else 4e-2
    assert (out.float() - ref).abs().max().item() < tol
    gg = torch.autograd.grad(out, ins, dH.to(dtype))
    for a_, r_ in zip(gg, rg):
        assert (a_.float() - r_).abs().max().item() < tol * max(1.0, r_.abs().max().item()) * (1 if dtype == torch.float32 else 4)
