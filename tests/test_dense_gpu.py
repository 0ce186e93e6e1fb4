"""Dense / output layers on the in-tree GEMM kernels (ops/dense.py): forward with bias epilogue, input gradient on the transposed
weight image, weight gradient on the split-K kernel — against the same network on the CPU (plain PyTorch fp32 math), with the library
GEMM entry points poisoned so a silent fallback fails the test."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mlp(dtype, dev, nIn=64, hidden=(256, 128), nOut=16):
    from deeplearning4j_amd.nd4j.learning import Sgd
    from deeplearning4j_amd.nn.conf.base import InputType
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import DenseLayer, OutputLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    b = NeuralNetConfiguration.Builder().seed(7).updater(Sgd(0.1)).weightInit("XAVIER")
    if dtype:
        b = b.computeDataType(dtype)
    lb = b.list()
    prev = nIn
    for h in hidden:
        lb = lb.layer(DenseLayer(nIn=prev, nOut=h, activation="TANH"))
        prev = h
    conf = lb.layer(OutputLayer(nIn=prev, nOut=nOut, activation="SOFTMAX", lossFunction="MCXENT")).setInputType(InputType.feedForward(nIn)).build()
    return MultiLayerNetwork(conf, device=dev).init()


@pytest.mark.parametrize("dtype", [None, "BFLOAT16", "HALF"])
def test_dense_network_runs_on_own_gemm_kernels(dtype, monkeypatch):
    a = _mlp(None, "cpu")
    b = _mlp(dtype, "cuda")
    assert torch.equal(a.params(), b.params().cpu())
    g = torch.Generator().manual_seed(0)
    x = torch.randn(96, 64, generator=g)
    y = torch.eye(16)[torch.randint(0, 16, (96,), generator=g)]
    a.computeGradientAndScore(x, y)

    def poisoned(*_a, **_k):
        raise AssertionError("library GEMM called for a shape the in-tree kernels support")
    import deeplearning4j_amd.nn.layers.core as CORE
    monkeypatch.setattr(CORE.F, "linear", poisoned)
    monkeypatch.setattr(CORE.torch, "matmul", poisoned)
    b.computeGradientAndScore(x.cuda(), y.cuda())
    monkeypatch.undo()
    tol = 1e-5 if dtype is None else (3e-2 if dtype == "BFLOAT16" else 4e-3)
    assert abs(float(a._score) - float(b._score)) < tol * max(1.0, abs(float(a._score)))
    ga, gb = a._grads.double(), b._grads.cpu().double()
    rel = float((ga - gb).norm() / ga.norm())
    assert rel < (2e-5 if dtype is None else tol), rel
    # and training moves both the same way (replica refresh of the transposed image included)
    for _ in range(5):
        a.fit(x, y)
        b.fit(x.cuda(), y.cuda())
    rel_p = float((a.params() - b.params().cpu()).norm() / a.params().norm())
    assert rel_p < (1e-5 if dtype is None else 2e-2), rel_p


def test_dense_odd_sizes_fall_back_and_match():
    a = _mlp(None, "cpu", nIn=30, hidden=(50,), nOut=3)
    b = _mlp(None, "cuda", nIn=30, hidden=(50,), nOut=3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(17, 30, generator=g)
    y = torch.eye(3)[torch.randint(0, 3, (17,), generator=g)]
    a.computeGradientAndScore(x, y)
    b.computeGradientAndScore(x.cuda(), y.cuda())
    assert float((a._grads - b._grads.cpu()).norm() / a._grads.norm()) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_samediff_linear_ops_run_on_own_gemm_kernels_forward_and_backward(dtype, monkeypatch):
    """xw_plus_b / matmul custom ops (the ops an imported BERT is made of): platform route on the in-tree kernels, differentiable,
    against fp32 PyTorch math; torch.matmul / addmm are poisoned inside the route."""
    from deeplearning4j_amd.nd4j.ops.custom import exec_op, OpExecutioner
    import deeplearning4j_amd.ops.linear as LIN
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(4, 50, 96, device="cuda", generator=g).to(dtype).reshape(200, 96)
    w = (torch.randn(96, 136, device="cuda", generator=g) / 10).to(dtype)
    bias = torch.randn(136, device="cuda", generator=g).to(dtype)
    ref_in = [v.float().detach().clone().requires_grad_(True) for v in (x, w, bias)]
    ref = ref_in[0] @ ref_in[1] + ref_in[2]
    gy = torch.randn_like(ref)
    rg = torch.autograd.grad(ref, ref_in, gy)
    OpExecutioner.resetHelperStatistics()
    ins = [v.detach().clone().requires_grad_(True) for v in (x, w, bias)]
    real_mm = torch.matmul
    monkeypatch.setattr(LIN.torch, "matmul", lambda *a, **k: (_ for _ in ()).throw(AssertionError("library GEMM in the linear route")))
    out = exec_op("xw_plus_b", ins)[0]
    got = torch.autograd.grad(out, ins, gy.to(dtype))
    monkeypatch.setattr(LIN.torch, "matmul", real_mm)
    assert OpExecutioner.helperStatistics()["hit"].get("xw_plus_b") == 1
    tol = 2e-4 if dtype == torch.float32 else 5e-2
    assert (out.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    for a_, r_ in zip(got, rg):
        assert (a_.float() - r_).abs().max().item() < tol * max(1.0, r_.abs().max().item())
    # matmul with a 3-D left operand and the batched NT form
    x3 = x.reshape(4, 50, 96)
    o3 = exec_op("matmul", [x3, w], iArgs=[0, 0, 0])[0]
    assert o3.shape == (4, 50, 136) and (o3.float() - x3.float() @ w.float()).abs().max().item() < tol * 10
    q = torch.randn(6, 40, 64, device="cuda", generator=g).to(dtype)
    k = torch.randn(6, 24, 64, device="cuda", generator=g).to(dtype)
    s = exec_op("matmul", [q, k], iArgs=[0, 1, 0], tArgs=[0.125, 0.0])[0]
    assert (s.float() - 0.125 * q.float() @ k.float().transpose(1, 2)).abs().max().item() < tol * 10
    assert OpExecutioner.helperStatistics()["hit"].get("matmul") == 2
    # a plain 2-D NN-form call without autograd is declined by the measured cost model and says so
    OpExecutioner.resetHelperStatistics()
    exec_op("matmul", [x.detach(), w.detach()])
    st = OpExecutioner.helperStatistics()
    assert not st["hit"].get("matmul") and any(k[0] == "matmul" and k[1].startswith("cost model") for k in st["fail"]), st
