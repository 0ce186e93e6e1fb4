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
