"""GPU numerics of the round-2 LDS-DMA core: implicit-GEMM convolution forward / data gradient (bf16, fp16, fp32; any stride, odd
channel counts, 1-D / 3-D) and the plain / batched NT GEMM with fused epilogues, each against a plain PyTorch fp32 reference of the
same op computed on the same (rounded) inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, C, H, W, K, R, stride, pad, dil
    (4, 64, 14, 14, 64, 1, 1, 0, 1),
    (3, 64, 13, 13, 128, 3, 1, 1, 1),
    (2, 64, 15, 15, 256, 1, 2, 0, 1),      # strided 1x1: residue classes without taps write zeros
    (2, 128, 8, 8, 128, 3, 2, 1, 1),       # strided 3x3 data gradient (library path in round 1)
    (2, 64, 12, 12, 64, 3, 1, 2, 2),       # dilation
    (2, 24, 9, 11, 40, 3, 1, 1, 1),        # channels not multiples of 64
    (3, 8, 10, 7, 72, 5, 3, 2, 1),         # 5x5 stride 3
    (16, 64, 28, 28, 64, 3, 1, 1, 1),
]
DTYPES = [torch.bfloat16, torch.float16, torch.float32]
TOL = {torch.bfloat16: 2e-2, torch.float16: 4e-3, torch.float32: 2e-5}


def _mk(N, C, H, W, K, R, dtype, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(N, C, H, W, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device="cuda", generator=g) / (C * R * R) ** 0.5).to(dtype)
    return x, w.contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES)
def test_forward_stats_all_tiles(case, dtype):
    from deeplearning4j_amd.ops import conv_mfma as CM
    N, C, H, W, K, R, s, p, d = case
    x, w = _mk(N, C, H, W, K, R, dtype)
    if not CM.supported(x, w):
        pytest.skip("channel count not a multiple of the 16-byte chunk for this dtype")
    b = torch.randn(K, device="cuda")
    ref = F.conv2d(x.float(), w.float(), b, s, p, d)
    for tile in (None, 0, 1, 2, 3):
        CM._FORCE_TILE = tile
        CM._DESC.clear()
        try:
            y, st = CM.forward(x, w, b, (s, s), (p, p), (d, d), want_stats=True)
        finally:
            CM._FORCE_TILE = None
        assert y.shape == ref.shape and y.dtype == dtype
        err = (y.float() - ref).abs().max().item()
        assert err < TOL[dtype] * max(1.0, ref.abs().max().item()), (tile, err)
        sums = st.sums()
        yf = y.float()
        rs, rq = yf.sum((0, 2, 3)), (yf * yf).sum((0, 2, 3))
        assert torch.allclose(sums[:K], rs, rtol=1e-3, atol=1e-4 * yf.abs().sum((0, 2, 3)).max().item()), tile
        assert torch.allclose(sums[K:], rq, rtol=1e-3, atol=1e-3 * rq.max().item()), tile
    CM._DESC.clear()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CASES)
def test_dgrad_any_stride_and_accumulate(case, dtype):
    from deeplearning4j_amd.ops import conv_mfma as CM
    N, C, H, W, K, R, s, p, d = case
    x, w = _mk(N, C, H, W, K, R, dtype, 1)
    if not CM.supported(x, w) or K % CM.chunk(dtype):
        pytest.skip("channel count not a multiple of the 16-byte chunk for this dtype")
    xf = x.float().requires_grad_(True)
    ref_y = F.conv2d(xf, w.float(), None, s, p, d)
    dz = torch.randn_like(ref_y).to(dtype).contiguous(memory_format=torch.channels_last)
    (ref_dx,) = torch.autograd.grad(ref_y, xf, dz.float())
    wd = CM.flip_filters(w)
    dx = CM.dgrad(dz, x.shape, wd, (s, s), (p, p), (d, d))
    assert dx.shape == x.shape and dx.is_contiguous(memory_format=torch.channels_last)
    assert not torch.isnan(dx.float()).any()
    err = (dx.float() - ref_dx).abs().max().item()
    assert err < TOL[dtype] * max(1.0, ref_dx.abs().max().item()), err
    prev = torch.randn(x.shape, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    want = (prev.float() + dx.float()).to(dtype)
    got = CM.dgrad(dz, x.shape, wd, (s, s), (p, p), (d, d), accum_into=prev)
    assert got.data_ptr() == prev.data_ptr()
    assert torch.equal(got, want)


def test_shifted_statistics_large_offset():
    """|mean| >> std: sums of (y - shift) keep the variance that E[y^2] - E[y]^2 in fp32 would lose."""
    from deeplearning4j_amd.ops import conv_mfma as CM
    N, C, H, W, K = 8, 64, 16, 16, 64
    x, w = _mk(N, C, H, W, K, 1, torch.float32)
    b = torch.full((K,), 3000.0, device="cuda")
    shift = torch.full((K,), 2999.0, device="cuda")
    y, st = CM.forward(x, w, b, (1, 1), (0, 0), (1, 1), want_stats=True, stat_shift=shift)
    sums = st.sums()
    M = N * H * W
    m1, m2 = sums[:K] / M, sums[K:] / M
    var = m2 - m1 * m1
    ref_var = y.double().var((0, 2, 3), unbiased=False)
    ref_mean = y.double().mean((0, 2, 3))
    assert torch.allclose((m1 + shift).double(), ref_mean, rtol=1e-6)
    assert torch.allclose(var.double(), ref_var, rtol=2e-3), (var[:4], ref_var[:4])


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv1d_and_conv3d(dtype):
    from deeplearning4j_amd.ops import conv_mfma as CM
    g = torch.Generator(device="cuda").manual_seed(3)
    x1 = torch.randn(4, 16, 37, device="cuda", generator=g).to(dtype).permute(0, 2, 1).contiguous().permute(0, 2, 1)
    w1 = (torch.randn(24, 16, 5, device="cuda", generator=g) / 9).to(dtype).permute(0, 2, 1).contiguous().permute(0, 2, 1)
    assert CM.supported(x1, w1)
    y1, _ = CM.forward(x1, w1, None, (2,), (2,), (1,))
    ref1 = F.conv1d(x1.float(), w1.float(), None, 2, 2, 1)
    assert (y1.float() - ref1).abs().max().item() < TOL[dtype] * max(1.0, ref1.abs().max().item())
    x3 = torch.randn(2, 8, 6, 7, 5, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last_3d)
    w3 = (torch.randn(16, 8, 3, 3, 3, device="cuda", generator=g) / 15).to(dtype).contiguous(memory_format=torch.channels_last_3d)
    assert CM.supported(x3, w3)
    y3, _ = CM.forward(x3, w3, None, (1, 2, 1), (1, 1, 0), (1, 1, 1))
    ref3 = F.conv3d(x3.float(), w3.float(), None, (1, 2, 1), (1, 1, 0), (1, 1, 1))
    assert y3.shape == ref3.shape
    assert (y3.float() - ref3).abs().max().item() < TOL[dtype] * max(1.0, ref3.abs().max().item())
    x3f = x3.float().requires_grad_(True)
    r = F.conv3d(x3f, w3.float(), None, (1, 2, 1), (1, 1, 0), (1, 1, 1))
    dz = torch.randn_like(r).to(dtype).contiguous(memory_format=torch.channels_last_3d)
    (rdx,) = torch.autograd.grad(r, x3f, dz.float())
    dx = CM.dgrad(dz, x3.shape, CM.flip_filters(w3), (1, 2, 1), (1, 1, 0), (1, 1, 1))
    assert (dx.float() - rdx).abs().max().item() < TOL[dtype] * max(1.0, rdx.abs().max().item())


GEMMS = [(300, 200, 136), (1024, 768, 768), (64, 64, 64), (4096, 3072, 768), (17, 1000, 264)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mnk", GEMMS)
def test_gemm_nt_epilogues(mnk, dtype):
    from deeplearning4j_amd.ops import gemm as G
    M, N, K = mnk
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    b = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g).to(dtype)
    base = a.float() @ b.float().t()
    tol = TOL[dtype] * max(1.0, base.abs().max().item())
    for tile in (None, 0, 1, 2, 3) + ((4,) if dtype == torch.bfloat16 else ()):
        y = G.gemm_nt(a, b, tile=tile)
        assert (y.float() - base).abs().max().item() < tol, tile
    y = G.gemm_nt(a, b, bias=bias, act="gelu")
    assert (y.float() - F.gelu(base + bias, approximate="tanh")).abs().max().item() < tol
    y = G.gemm_nt(a, b, bias=bias, act="relu", residual=res)
    assert (y.float() - (F.relu(base + bias) + res.float())).abs().max().item() < 2 * tol
    y = G.gemm_nt(a, b, alpha=0.5, out_dtype=torch.float32)
    assert y.dtype == torch.float32 and (y - 0.5 * base).abs().max().item() < tol


def test_gemm_batched_strided():
    from deeplearning4j_amd.ops import gemm as G
    g = torch.Generator(device="cuda").manual_seed(6)
    q = torch.randn(6, 128, 64, device="cuda", generator=g).bfloat16()
    k = torch.randn(6, 96, 64, device="cuda", generator=g).bfloat16()
    y = G.bgemm_nt(q, k, alpha=0.125)
    ref = 0.125 * torch.einsum("bmk,bnk->bmn", q.float(), k.float())
    assert (y.float() - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())


WG_CASES = [(4, 64, 14, 14, 64, 1, 1, 0, 1), (3, 64, 13, 13, 128, 3, 1, 1, 1), (2, 24, 9, 11, 40, 3, 1, 1, 1),
            (2, 128, 8, 8, 128, 3, 2, 1, 1), (3, 8, 10, 7, 72, 5, 3, 2, 1), (2, 2048, 4, 4, 1000, 1, 1, 0, 1)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", WG_CASES)
def test_wgrad_all_dtypes_and_partial_channel_tiles(case, dtype):
    """Weight gradient on the own kernels for bf16 / fp16 (transposing-read MFMA) and fp32 (exact fp32 MFMA), including channel counts
    that are not multiples of the 64-wide tiles (masked loads / stores)."""
    from deeplearning4j_amd.ops import conv_igemm as IG
    N, C, H, W, K, R, s, p, d = case
    x, w = _mk(N, C, H, W, K, R, dtype, 2)
    assert IG.wgrad_supported(x, w)
    wf = w.float().requires_grad_(True)
    ref_y = F.conv2d(x.float(), wf, None, s, p, d)
    dz = torch.randn_like(ref_y).to(dtype).contiguous(memory_format=torch.channels_last)
    (ref_dw,) = torch.autograd.grad(ref_y, wf, dz.float())
    dw = IG.wgrad(dz, x, w.shape, (s, s), (p, p), (d, d))
    assert dw.shape == ref_dw.shape and dw.dtype == torch.float32
    tol = {torch.bfloat16: 1e-4, torch.float16: 1e-4, torch.float32: 2e-5}[dtype]      # fp32 accumulation of exact products
    err = (dw - ref_dw).abs().max().item()
    assert err < tol * max(1.0, ref_dw.abs().max().item()), err
    out = torch.zeros_like(ref_dw)
    assert IG.wgrad(dz, x, w.shape, (s, s), (p, p), (d, d), dw_out=out, prezeroed=True) is None
    assert (out - ref_dw).abs().max().item() < tol * max(1.0, ref_dw.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_conv_layer_dispatch_uses_own_kernels(dtype, monkeypatch):
    """conv2d_forward / conv2d_backward in fp32 and fp16 must not reach the library: aten convolution entry points are poisoned."""
    from deeplearning4j_amd.ops import conv as CONV
    x, w = _mk(4, 32, 12, 12, 48, 3, dtype, 7)
    ref = F.conv2d(x.float(), w.float(), None, 2, 1, 1)
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    ry = F.conv2d(xf, wf, None, 2, 1, 1)
    dz = torch.randn_like(ry).to(dtype).contiguous(memory_format=torch.channels_last)
    rdx, rdw = torch.autograd.grad(ry, [xf, wf], dz.float())

    def boom(*a, **k):
        raise AssertionError("library convolution called")
    monkeypatch.setattr(F, "conv2d", boom)
    monkeypatch.setattr(torch.ops.aten, "convolution_backward", boom, raising=False)
    y, _ = CONV.conv2d_forward(x, w, None, (2, 2), (1, 1), (1, 1))
    dx, dw, _ = CONV.conv2d_backward(dz, x, w, (2, 2), (1, 1), (1, 1))
    t = TOL[dtype]
    assert (y.float() - ref).abs().max().item() < t * max(1.0, ref.abs().max().item())
    assert (dx.float() - rdx).abs().max().item() < t * max(1.0, rdx.abs().max().item())
    assert (dw.float() - rdw).abs().max().item() < 1e-3 * max(1.0, rdw.abs().max().item())
