"""Fused SDPA forward kernel (native/csrc/attention.hip) vs a plain PyTorch fp32 reference; backward through the autograd wrapper."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, mask, scale, causal):
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    if causal:
        Tq, Tk = s.shape[-2:]
        s = s.masked_fill(~torch.ones(Tq, Tk, dtype=torch.bool, device=s.device).tril(), float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    return torch.matmul(p, v.float())


@pytest.mark.parametrize("B,H,Tq,Tk,D,causal,masked", [
    (2, 3, 128, 128, 64, False, False), (2, 2, 200, 333, 64, False, True), (1, 4, 77, 77, 64, True, False),
    (2, 2, 256, 256, 128, False, True), (1, 2, 130, 130, 128, True, True), (3, 12, 128, 128, 64, False, True)])
def test_sdpa_forward_matches_reference(B, H, Tq, Tk, D, causal, masked):
    from deeplearning4j_amd.ops import attention as A
    torch.manual_seed(B * 1000 + Tq + D)
    dev = torch.device("cuda")
    q, k, v = (torch.randn(B, H, T, D, device=dev).bfloat16() for T in (Tq, Tk, Tk))
    mask = None
    if masked:
        lens = torch.randint(Tk // 2, Tk + 1, (B,), device=dev)
        mask = (torch.arange(Tk, device=dev)[None, :] < lens[:, None]).float()
    scale = 1.0 / math.sqrt(D)
    assert A._hip_ok(q, k, v, mask, 0.0, False)
    o = A.sdpa(q, k, v, mask, scale, causal)
    r = _ref(q, k, v, mask, scale, causal)
    err = (o.float() - r).abs().max().item()
    assert err < 3e-2, err
    assert torch.isfinite(o.float()).all()


def test_sdpa_fully_masked_rows_and_backward():
    from deeplearning4j_amd.ops import attention as A
    torch.manual_seed(5)
    dev = torch.device("cuda")
    B, H, T, D = 2, 2, 96, 64
    q, k, v = (torch.randn(B, H, T, D, device=dev).bfloat16().requires_grad_(True) for _ in range(3))
    mask = torch.ones(B, T, device=dev)
    mask[1, 40:] = 0
    scale = 1.0 / math.sqrt(D)
    o = A.sdpa(q, k, v, mask, scale, False)
    g = torch.randn_like(o)
    o.backward(g)
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    r = _ref(qf, kf, vf, mask, scale, False)
    r.backward(g.float())
    for a, b_, name in ((q.grad, qf.grad, "dq"), (k.grad, kf.grad, "dk"), (v.grad, vf.grad, "dv")):
        rel = (a.float() - b_).norm() / b_.norm().clamp_min(1e-6)
        assert rel < 3e-2, (name, float(rel))
    # a batch element whose keys are ALL masked yields zeros, not NaN
    mask0 = torch.ones(B, T, device=dev)
    mask0[0] = 0
    o0 = A.sdpa(q.detach(), k.detach(), v.detach(), mask0, scale, False)
    assert torch.isfinite(o0.float()).all() and float(o0[0].abs().max()) == 0.0


@pytest.mark.parametrize("B,H,Tq,Tk,D,causal,masked", [
    (2, 3, 128, 128, 64, False, False),
    (1, 2, 200, 200, 64, True, False),          # ragged tiles + causal
    (2, 2, 77, 150, 64, False, True),           # cross attention lengths, key padding mask
    (1, 2, 256, 256, 128, False, False),
    (2, 1, 130, 130, 128, True, True),
    (1, 9, 64, 320, 128, False, True),          # more (batch, head) pairs than XCDs
])
def test_sdpa_fused_backward_matches_fp32_reference(B, H, Tq, Tk, D, causal, masked):
    """dq / dk / dv of the fused backward kernels vs autograd through a plain fp32 softmax(QK^T)V."""
    from deeplearning4j_amd.ops import attention as A
    torch.manual_seed(B * 1000 + Tq + Tk + D)
    dev = torch.device("cuda")
    q = torch.randn(B, H, Tq, D, device=dev).bfloat16().requires_grad_(True)
    k = torch.randn(B, H, Tk, D, device=dev).bfloat16().requires_grad_(True)
    v = torch.randn(B, H, Tk, D, device=dev).bfloat16().requires_grad_(True)
    mask = None
    if masked:
        lens = torch.randint(Tk // 2, Tk + 1, (B,), device=dev)
        mask = (torch.arange(Tk, device=dev)[None, :] < lens[:, None]).float()
    scale = 1.0 / math.sqrt(D)
    o = A.sdpa(q, k, v, mask, scale, causal)
    g = torch.randn_like(o)
    o.backward(g)
    qf, kf, vf = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    r = _ref(qf, kf, vf, mask, scale, causal)
    r.backward(g.float())
    for a, b_, name in ((q.grad, qf.grad, "dq"), (k.grad, kf.grad, "dk"), (v.grad, vf.grad, "dv")):
        assert torch.isfinite(a.float()).all(), name
        rel = (a.float() - b_).norm() / b_.norm().clamp_min(1e-6)
        assert rel < 2e-2, (name, float(rel))
        # element-wise too: no mis-mapped rows hiding inside a small relative norm
        assert float((a.float() - b_).abs().max()) < 0.05 * float(b_.abs().max()) + 1e-3, name


def test_self_attention_network_uses_flash_kernel_and_trains():
    """SelfAttentionLayer (multi-head, head size 64) in a bf16 network: the layer's SDPA goes through the HIP kernel."""
    from deeplearning4j_amd.nd4j.learning import Adam
    from deeplearning4j_amd.nn.conf.base import InputType
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import SelfAttentionLayer, GlobalPoolingLayer, OutputLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    from deeplearning4j_amd.ops import attention as A
    calls = []
    orig = A._FlashSDPA.apply

    def spy(*a):
        calls.append(a[0].shape)
        return orig(*a)
    A._FlashSDPA.apply = spy
    try:
        conf = (NeuralNetConfiguration.Builder().seed(1).updater(Adam(2e-3)).weightInit("XAVIER").computeDataType("BFLOAT16").list()
                .layer(SelfAttentionLayer(nIn=32, nOut=64, nHeads=2, headSize=64, projectInput=True))
                .layer(GlobalPoolingLayer(poolingType="AVG"))
                .layer(OutputLayer(nIn=64, nOut=3, activation="SOFTMAX", lossFunction="MCXENT"))
                .setInputType(InputType.recurrent(32, 48)).build())
        net = MultiLayerNetwork(conf, device="cuda").init()
        torch.manual_seed(0)
        x = torch.randn(24, 32, 48, device="cuda")
        cls = torch.arange(24, device="cuda") % 3
        x[:, :3, :] += torch.nn.functional.one_hot(cls, 3).float().unsqueeze(2) * 2.0
        y = torch.nn.functional.one_hot(cls, 3).float()
        net.fit(x, y)
        s0 = net.score()
        for _ in range(80):
            net.fit(x, y)
        assert calls and calls[0][1] == 2 and calls[0][3] == 64, "flash SDPA kernel was not used by the attention layer"
        assert net.score() < 0.5 * s0, (s0, net.score())
    finally:
        A._FlashSDPA.apply = orig
