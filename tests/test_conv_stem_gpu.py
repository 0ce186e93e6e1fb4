"""3-channel stem convolution kernels (native/csrc/conv_stem.hip) vs F.conv2d in fp32."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W,pad", [(3, 224, 224, 3), (2, 97, 131, 3), (1, 64, 250, 0), (5, 33, 40, 2)])
def test_stem_forward_stats_and_wgrad(N, H, W, pad):
    from deeplearning4j_amd.ops import conv_stem as CS
    torch.manual_seed(N * 100 + H)
    dev = torch.device("cuda")
    x = torch.randn(N, 3, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, device=dev) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(64, device=dev)
    assert CS.supported(x, w, (2, 2), (pad, pad), (1, 1))
    y, st = CS.forward(x, w, bias, (2, 2), (pad, pad), (1, 1), want_stats=True)
    ref = F.conv2d(x.float(), w.float(), bias, 2, pad)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(y.float(), ref, atol=3e-2, rtol=2e-2), float((y.float() - ref).abs().max())
    sums = st.sums()
    yf = y.float()
    assert torch.allclose(sums[:64], yf.sum((0, 2, 3)), rtol=1e-3, atol=1e-1 + 1e-3 * float(yf.abs().sum((0, 2, 3)).max()))
    assert torch.allclose(sums[64:], (yf * yf).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    y2, none = CS.forward(x, w, None, (2, 2), (pad, pad), (1, 1), want_stats=False)
    assert none is None and torch.allclose(y2.float(), ref - bias.reshape(1, -1, 1, 1), atol=3e-2, rtol=2e-2)

    dz = torch.randn_like(ref).bfloat16().contiguous(memory_format=torch.channels_last)
    dw = CS.wgrad(dz, x, (2, 2), (pad, pad), (1, 1))
    dw_ref = torch.ops.aten.convolution_backward(dz.float(), x.float(), w.float(), None, [2, 2], [pad, pad], [1, 1], False, [0, 0], 1,
                                                 [False, True, False])[1]
    rel = (dw - dw_ref).norm() / dw_ref.norm()
    assert rel < 5e-3, float(rel)
    # second call: the scratch was left zeroed, writing into a caller-provided gradient view
    out = torch.full((64 * 147,), 7.0, device=dev)
    assert CS.wgrad(dz, x, (2, 2), (pad, pad), (1, 1), dw_out=out) is None
    assert torch.allclose(out.reshape(64, 3, 7, 7), dw, rtol=1e-4, atol=1e-3)
