"""execCustomOp2 with DEVICE arrays (special buffers): the native executors launch the hand-written gfx950 kernels of libdl4j_hip.so
directly from C++ (no Python op code in between); results against plain PyTorch fp32 math."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_exec_custom_op2_device_arrays(dt):
    from deeplearning4j_amd.nd4j import native_ops as N
    g = torch.Generator(device="cuda").manual_seed(0)
    tol = 1e-5 if dt == torch.float32 else 2e-2
    a = torch.randn(33, 1, 64, device="cuda", generator=g).to(dt)
    b = torch.randn(17, 1, device="cuda", generator=g).to(dt)
    out = N.exec_custom_op("add", [a, b])[0]
    assert out.is_cuda and torch.allclose(out.float(), (a.float() + b.float()).to(dt).float(), atol=tol)
    assert torch.allclose(N.exec_custom_op("multiply", [a, a])[0].float(), (a.float() * a.float()), atol=tol, rtol=tol)
    x = torch.randn(300, 264, device="cuda", generator=g).to(dt)
    assert torch.allclose(N.exec_custom_op("relu", [x], tArgs=[0.0])[0].float(), x.float().clamp_min(0))
    assert torch.allclose(N.exec_custom_op("tanh", [x])[0].float(), torch.tanh(x.float()), atol=tol)
    assert torch.allclose(N.exec_custom_op("softmax", [x])[0].float(), torch.softmax(x.float(), -1), atol=tol)
    w = (torch.randn(200, 264, device="cuda", generator=g) / 16).to(dt)
    ref = x.float() @ w.float().t()
    got = N.exec_custom_op("matmul", [x, w], iArgs=[0, 1, 0])[0]
    assert got.dtype == dt and (got.float() - ref).abs().max().item() < (1e-4 if dt == torch.float32 else 5e-2) * max(1.0, ref.abs().max().item())
    bias = torch.randn(264, device="cuda", generator=g)
    assert torch.allclose(N.exec_custom_op("biasadd", [x, bias])[0].float(), (x.float() + bias).to(dt).float(), atol=tol)
    torch.cuda.synchronize()
