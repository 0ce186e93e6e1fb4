"""GPU numerics: implicit-GEMM MFMA conv kernels (forward + fused BN-stat epilogue, dgrad, wgrad)
vs a plain PyTorch fp32 reference of the same op (reference computed on the bf16-rounded inputs)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, C, H, W, K, R, stride, pad, dil
    (4, 64, 14, 14, 64, 1, 1, 0, 1),
    (3, 64, 13, 13, 128, 3, 1, 1, 1),
    (2, 128, 9, 11, 64, 3, 1, 1, 1),
    (2, 64, 15, 15, 256, 1, 2, 0, 1),
    (5, 256, 7, 7, 128, 1, 1, 0, 1),
    (2, 64, 12, 12, 64, 3, 1, 2, 2),
    (2, 128, 8, 8, 128, 3, 2, 1, 1),
    (16, 64, 28, 28, 64, 3, 1, 1, 1),
    # network-scale launches: the 128-row / 128-column tile instantiations and their fused data-gradient epilogues are only chosen
    # from these sizes up (ResNet-50 at batch 256: the first of them once ran with a loader that the small cases never reached)
    (256, 256, 7, 7, 256, 3, 1, 1, 1),
    (32, 128, 28, 28, 128, 3, 1, 1, 1),
    (64, 1024, 7, 7, 256, 1, 1, 0, 1),
    (16, 256, 28, 28, 512, 1, 2, 0, 1),
]


def _mk(N, C, H, W, K, R, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(N, C, H, W, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device="cuda", generator=g) / (C * R * R) ** 0.5).bfloat16()
    w = w.contiguous(memory_format=torch.channels_last)   # OHWI memory
    return x, w


@pytest.mark.parametrize("case", CASES)
def test_forward_and_stats(case):
    from deeplearning4j_amd.ops import conv_igemm as IG
    N, C, H, W, K, R, s, p, d = case
    x, w = _mk(N, C, H, W, K, R)
    b = torch.randn(K, device="cuda")
    assert IG.supported(x, w, (s, s), (p, p), (d, d))
    y, st = IG.forward(x, w, b, (s, s), (p, p), (d, d), want_stats=True)
    sums = st.sums()
    ref = F.conv2d(x.float(), w.float(), b, s, p, d)
    assert y.shape == ref.shape
    err = (y.float() - ref).abs().max().item()
    assert err < 2e-2 * max(1.0, ref.abs().max().item()), err
    yf = y.float()
    rs = yf.sum((0, 2, 3))
    rq = (yf * yf).sum((0, 2, 3))
    assert torch.allclose(sums[:K], rs, rtol=1e-3, atol=1e-2 * yf.abs().sum((0, 2, 3)).max().item() / 100)
    assert torch.allclose(sums[K:], rq, rtol=1e-3, atol=1e-3 * rq.max().item())


def test_forward_statistics_pivot_survives_large_mean():
    """Statistics epilogue with a pivot (the consuming BatchNorm's running mean): outputs with |mean| >> std, where the unshifted
    E[y^2] - E[y]^2 of fp32 partial sums loses the variance to cancellation. Reference: fp64 statistics of the stored bf16 outputs."""
    from deeplearning4j_amd.ops import conv_igemm as IG
    from deeplearning4j_amd.ops import batchnorm as BN
    g = torch.Generator(device="cuda").manual_seed(7)
    N, C, H, K = 16, 64, 28, 64
    x = (1.0 + 0.1 * torch.randn(N, C, H, H, device="cuda", generator=g)).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (100.0 / C + 0.02 * torch.randn(K, C, 1, 1, device="cuda", generator=g)).bfloat16().contiguous(memory_format=torch.channels_last)
    M = N * H * H
    gamma, beta = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
    gvar = torch.ones(K, device="cuda")
    res = {}
    for name, gmean in (("pivot", torch.full((K,), 100.0, device="cuda")), ("plain", None)):
        y, st = IG.forward(x, w, None, (1, 1), (0, 0), (1, 1), want_stats=True, stat_pivot=None if gmean is None else (gmean, None))
        assert (st.pivot is not None) == (gmean is not None)
        y2 = y.permute(0, 2, 3, 1).reshape(M, K)
        dgm, dgv = torch.empty(K, device="cuda"), torch.empty(K, device="cuda")
        gm = gmean if gmean is not None else torch.zeros(K, device="cuda")
        mean, invstd, _, _ = BN.batch_stats(y2, gamma, beta, 1e-5, 0.9, gm, gvar, False, None, dgm, dgv, sums=st)
        yd = y2.double()
        ref_mean, ref_var = yd.mean(0), yd.var(0, unbiased=False)
        assert ref_mean.abs().min().item() > 50 * ref_var.sqrt().max().item()          # the regime this test is about
        var = 1.0 / invstd.double() ** 2 - 1e-5
        res[name] = ((mean.double() - ref_mean).abs().max().item(), ((var - ref_var).abs() / ref_var).max().item())
    assert res["pivot"][0] < 1e-3 and res["pivot"][1] < 1e-3, res
    assert res["plain"][1] > 5 * res["pivot"][1], res            # the cancellation the pivot removes


@pytest.mark.parametrize("case", CASES)
def test_dgrad(case):
    from deeplearning4j_amd.ops import conv_igemm as IG
    N, C, H, W, K, R, s, p, d = case
    x, w = _mk(N, C, H, W, K, R, 1)
    if not IG.dgrad_supported(w, (s, s), (p, p)):
        pytest.skip("strided non-1x1 dgrad uses the library path")
    xf = x.float().requires_grad_(True)
    ref_y = F.conv2d(xf, w.float(), None, s, p, d)
    dz = torch.randn_like(ref_y).bfloat16().contiguous(memory_format=torch.channels_last)
    (ref_dx,) = torch.autograd.grad(ref_y, xf, dz.float())
    w_dgrad = torch.flip(w, dims=[2, 3]).permute(1, 2, 3, 0).contiguous()   # [C][R][S][K]
    dx = IG.dgrad(dz, x.shape, w_dgrad, (s, s), (p, p), (d, d))
    err = (dx.float() - ref_dx).abs().max().item()
    assert err < 2e-2 * max(1.0, ref_dx.abs().max().item()), err
    # accumulate epilogue: an existing gradient for the same tensor is updated in place (graph fan-out points)
    prev = torch.randn(x.shape, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    assert IG.accum_ok(prev, x.shape)
    want = (prev.float() + dx.float()).bfloat16()
    got = IG.dgrad(dz, x.shape, w_dgrad, (s, s), (p, p), (d, d), accum_into=prev)
    assert got.data_ptr() == prev.data_ptr()
    assert torch.equal(got, want)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mask", ["bits", "coeffs", "none"])
def test_dgrad_fused_epilogue_masked_accumulate_and_bn_backward_sums(case, mask):
    """dgrad epilogue (EPI == 1): adds a gradient through ReLU mask bits without it being materialised and emits the partial sums
    of the BatchNorm backward reductions of the layer its output flows into; both against plain fp32 PyTorch on the same tensors."""
    from deeplearning4j_amd.ops import conv_igemm as IG
    from deeplearning4j_amd.ops import batchnorm as BN
    N, C, H, W, K, R, s, p, d = case
    x, w = _mk(N, C, H, W, K, R, 3)
    if not IG.dgrad_supported(w, (s, s), (p, p)):
        pytest.skip("strided non-1x1 dgrad uses the library path")
    g = torch.Generator(device="cuda").manual_seed(11)
    OH = (H + 2 * p - d * (R - 1) - 1) // s + 1
    dz = torch.randn(N, K, OH, OH if H == W else (W + 2 * p - d * (R - 1) - 1) // s + 1, device="cuda", generator=g).bfloat16()
    dz = dz.contiguous(memory_format=torch.channels_last)
    w_dgrad = torch.flip(w, dims=[2, 3]).permute(1, 2, 3, 0).contiguous()
    base = IG.dgrad(dz, x.shape, w_dgrad, (s, s), (p, p), (d, d))                   # plain kernel
    M = N * H * W
    # the BatchNorm this gradient flows into: input y (|mean| >> std on some channels), mean / invstd, ReLU mask
    y = (torch.randn(M, C, device="cuda", generator=g) * 0.5 + torch.linspace(-20, 20, C, device="cuda")).bfloat16()
    mean = y.float().mean(0)
    invstd = torch.rsqrt(y.float().var(0, unbiased=False) + 1e-5)
    ms = torch.rand(C, device="cuda", generator=g) + 0.5
    mt = -mean * ms + torch.randn(C, device="cuda", generator=g) * 0.1
    mbits = (torch.rand(M, C, device="cuda", generator=g) > 0.4)
    packed = (mbits.view(M, C // 8, 8).to(torch.uint8) << torch.arange(8, device="cuda", dtype=torch.uint8)).sum(-1).to(torch.uint8)
    sink = IG.BnBackwardSink(y, mean, invstd, ms if mask == "coeffs" else None, mt if mask == "coeffs" else None,
                             packed.reshape(-1) if mask == "bits" else None)
    assert sink.ok(x.shape)
    # the gradient added through mask bits (stride-1 launches only: a scatter launch skips rows)
    acc = None
    if s == 1:
        src = torch.randn(x.shape, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        abits = (torch.rand(M, C, device="cuda", generator=g) > 0.5)
        apacked = (abits.view(M, C // 8, 8).to(torch.uint8) << torch.arange(8, device="cuda", dtype=torch.uint8)).sum(-1).to(torch.uint8)
        acc = IG.MaskedGrad(src, apacked.reshape(-1))
        assert IG.MaskedGrad.ok(acc.src, acc.bits)
        want_mat = torch.where(abits.view(N, H, W, C), src.permute(0, 2, 3, 1), torch.zeros((), device="cuda", dtype=torch.bfloat16))
        assert torch.equal(acc.materialize().permute(0, 2, 3, 1), want_mat)
    dx = IG.dgrad(dz, x.shape, w_dgrad, (s, s), (p, p), (d, d), acc_masked=acc, bn_sink=sink)
    want = base if acc is None else (base.float() + acc.materialize().float()).bfloat16()
    assert torch.equal(dx, want)
    assert sink.partial is not None and sink.gy > 0
    # reference sums in fp32 from the tensor the kernel stored
    d2 = dx.permute(0, 2, 3, 1).reshape(M, C).float()
    if mask == "bits":
        d2 = d2 * mbits
    elif mask == "coeffs":
        d2 = d2 * ((y.float() * ms + mt) > 0)
    r0 = d2.sum(0)
    r1 = (d2 * (y.float() - mean) * invstd).sum(0)
    got = sink.partial.view(sink.gy, 2, C).sum(0)
    scale0 = d2.abs().sum(0).max().item()
    assert torch.allclose(got[0], r0, rtol=1e-4, atol=1e-5 * scale0), (got[0] - r0).abs().max()
    scale1 = (d2 * (y.float() - mean) * invstd).abs().sum(0).max().item()
    assert torch.allclose(got[1], r1, rtol=1e-4, atol=1e-5 * scale1), (got[1] - r1).abs().max()
    # and through the BatchNorm backward: identical dx / dgamma / dbeta to the column-reduction form on the same dy
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    dy2 = dx.permute(0, 2, 3, 1).reshape(M, C)
    kw = dict(mask_bits=packed.reshape(-1)) if mask == "bits" else (dict(mask_coeffs=(ms, mt)) if mask == "coeffs" else {})
    dg1, db1, dg2, db2 = (torch.empty(C, device="cuda") for _ in range(4))
    a, _ = BN.backward(y, dy2, None, mean, invstd, gamma, dg1, db1, **kw)
    b, _ = BN.backward(y, dy2, None, mean, invstd, gamma, dg2, db2, pre_partials=(sink.partial, sink.gy), **kw)
    assert torch.allclose(dg1, dg2, rtol=1e-4, atol=1e-5 * scale1) and torch.allclose(db1, db2, rtol=1e-4, atol=1e-5 * scale0)
    assert (a.float() - b.float()).abs().max().item() <= 2e-2 * max(1.0, a.float().abs().max().item())


@pytest.mark.parametrize("case", CASES)
def test_wgrad(case):
    from deeplearning4j_amd.ops import conv_igemm as IG
    N, C, H, W, K, R, s, p, d = case
    x, w = _mk(N, C, H, W, K, R, 2)
    wf = w.float().requires_grad_(True)
    ref_y = F.conv2d(x.float(), wf, None, s, p, d)
    dz = torch.randn_like(ref_y).bfloat16().contiguous(memory_format=torch.channels_last)
    (ref_dw,) = torch.autograd.grad(ref_y, wf, dz.float())
    for tb in (1, 64):    # no split (plain stores) and split-K (atomics)
        out = torch.full((K, C, R, R), float("nan"), device="cuda")
        assert IG.wgrad(dz, x, w.shape, (s, s), (p, p), (d, d), dw_out=out, target_blocks=tb) is None
        err = (out - ref_dw).abs().max().item()
        assert err < 1e-2 * max(1.0, ref_dw.abs().max().item()), (tb, err)


def test_cast_table_dgrad_layout():
    from deeplearning4j_amd.ops import updater as U
    O, I, kh, kw = 6, 5, 3, 3
    n = O * I * kh * kw
    src = torch.randn(n, device="cuda")
    ct = U.CastTable([(0, 0, n, I, kh * kw, O, 1)], "cuda")
    dst = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    ct.run(src, dst)
    w = src.view(O, I, kh, kw)
    ref = torch.flip(w, dims=[2, 3]).permute(1, 2, 3, 0).contiguous().bfloat16()
    assert torch.equal(dst.view(I, kh, kw, O), ref)


@pytest.mark.parametrize("O,I,K", [(64, 64, 9), (512, 512, 9), (256, 64, 1), (2048, 512, 1), (64, 3, 49), (128, 256, 1), (96, 40, 25)])
def test_cast_table_lds_staged_permutes_at_network_sizes(O, I, K):
    """Both permuting modes over many 2048-element chunks (LDS-staged path, and the direct fallback when a chunk's sources do
    not fit), with odd destination offsets as in the real arena."""
    from deeplearning4j_amd.ops import updater as U
    n = O * I * K
    torch.manual_seed(O + I + K)
    src = torch.randn(7 + n, device="cuda")
    ct = U.CastTable([(7, 3, n, I, K, 0, 0), (7, 3 + n + 5, n, I, K, O, 1)], "cuda")
    dst = torch.zeros(3 + 2 * n + 5, dtype=torch.bfloat16, device="cuda")
    ct.run(src, dst)
    w = src[7:].view(O, I, K)
    assert torch.equal(dst[3:3 + n].view(O, K, I), w.permute(0, 2, 1).contiguous().bfloat16())
    assert torch.equal(dst[3 + n + 5:].view(I, K, O), torch.flip(w, dims=[2]).permute(1, 2, 0).contiguous().bfloat16())
    assert float(dst[:3].abs().sum()) == 0 and float(dst[3 + n:3 + n + 5].abs().sum()) == 0


@pytest.mark.parametrize("O,I,K,tail", [(64, 64, 9, 0), (512, 512, 9, 0), (256, 64, 1, 0), (1000, 2048, 1, 3), (64, 3, 49, 0), (130, 36, 9, 0), (31, 7, 1, 2)])
def test_cast_table_vector_paths_with_aligned_offsets(O, I, K, tail):
    """16-byte aligned entries (what the real arena holds for the convolution / dense weights): the vector load / 4-element store
    paths of the straight cast and of the OIHW -> OHWI permute, including a scalar tail and shapes that fall back (I % 4 != 0)."""
    from deeplearning4j_amd.ops import updater as U
    n = O * I * K
    torch.manual_seed(O + I + K)
    src = torch.randn(8 + n + tail + 4, device="cuda")
    n2 = n + tail if K == 1 else n
    ct = U.CastTable([(8, 4, n2, I if K > 1 else n2, K, 0, 0)], "cuda")
    dst = torch.zeros(4 + n2 + 8, dtype=torch.bfloat16, device="cuda")
    ct.run(src, dst)
    if K == 1:
        assert torch.equal(dst[4:4 + n2], src[8:8 + n2].bfloat16())
    else:
        assert torch.equal(dst[4:4 + n].view(O, K, I), src[8:8 + n].view(O, I, K).permute(0, 2, 1).contiguous().bfloat16())
    assert float(dst[:4].abs().sum()) == 0 and float(dst[4 + n2:].abs().sum()) == 0
    # the flipped [I, K, O] image for the data-gradient kernels
    ct2 = U.CastTable([(8, 4, n, I, K, O, 1)], "cuda")
    dst2 = torch.zeros(4 + n + 8, dtype=torch.bfloat16, device="cuda")
    ct2.run(src, dst2)
    w = src[8:8 + n].view(O, I, K)
    assert torch.equal(dst2[4:4 + n].view(I, K, O), torch.flip(w, dims=[2]).permute(1, 2, 0).contiguous().bfloat16())
    assert float(dst2[:4].abs().sum()) == 0 and float(dst2[4 + n:].abs().sum()) == 0


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (1000, 192, 320), (4096, 768, 768)])
def test_hand_written_gemm_nt_matches_fp32(M, N, K):
    from deeplearning4j_amd.ops import gemm as G
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    b = torch.randn(N, device="cuda")
    ref = x.float() @ w.float().t() + b
    got = G.gemm_nt(x, w, b).float()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) < 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("shape", [(8, 64, 28, 28, 256, 1), (4, 128, 14, 14, 128, 3), (64, 64, 28, 28, 64, 1), (2, 256, 7, 7, 1024, 1),
                                   (32, 64, 56, 56, 64, 1)])
@pytest.mark.parametrize("log_std", [False, True])
def test_conv_epilogue_runs_the_batchnorm_finalize(shape, log_std, monkeypatch):
    """The consuming BatchNorm's mean / invstd / scale / shift and running-statistic deltas come out of the conv launch itself
    (last-workgroup finalize, native/csrc/ticket.h) and equal the statistics of the rounded conv output computed by PyTorch in fp32;
    repeated launches reuse the self-resetting counters."""
    from deeplearning4j_amd.ops import conv_igemm as IG
    from deeplearning4j_amd.ops import batchnorm as BN
    N, C, H, W, K, R = shape
    monkeypatch.setattr(BN, "_FUSE_STAGE2", True)
    x, w = _mk(N, C, H, W, K, R, seed=3)
    g = torch.Generator(device="cuda").manual_seed(5)
    gamma = torch.rand(K, device="cuda", generator=g) + 0.5
    beta = torch.randn(K, device="cuda", generator=g)
    cb = torch.randn(K, device="cuda", generator=g)
    gmean = torch.randn(K, device="cuda", generator=g)
    gvar = torch.rand(K, device="cuda", generator=g) * (0.3 if log_std else 1.0) + 0.1
    eps, decay = 1e-5, 0.9
    for rep in range(4):
        dgm, dgv = torch.full((K,), 7.0, device="cuda"), torch.full((K,), 7.0, device="cuda")
        req = BN.FinalizeRequest(gamma, beta, cb, gmean, gvar, dgm, dgv, eps, decay, log_std)
        y, fin = IG.forward(x, w, None, (1, 1), (R // 2, R // 2), (1, 1), want_stats=True, bn_fin=req)
        assert isinstance(fin, BN.Finalized)
        yf = y.float().permute(0, 2, 3, 1).reshape(-1, K)
        mean, var = yf.mean(0), yf.var(0, unbiased=False)
        assert torch.allclose(fin.mean, mean, rtol=1e-4, atol=1e-5)
        assert torch.allclose(fin.invstd, torch.rsqrt(var + eps), rtol=2e-3)
        assert torch.allclose(fin.scale, gamma * torch.rsqrt(var + eps), rtol=2e-3)
        assert torch.allclose(fin.shift, beta - mean * gamma * torch.rsqrt(var + eps), rtol=2e-3, atol=1e-3)
        rm, rv = BN.running_stat_grads(mean, var, gmean, gvar, decay, log_std, cb)
        assert torch.allclose(dgm, rm, rtol=1e-4, atol=1e-5)
        assert torch.allclose(dgv, rv, rtol=2e-3, atol=1e-4)
    cnt, _ = BN.ticket_ws(x.device)
    assert int(cnt.abs().sum()) == 0                       # every counter was handed back at zero


def test_bn_backward_fused_second_stage_equals_two_launch_form():
    from deeplearning4j_amd.ops import batchnorm as BN
    g = torch.Generator(device="cuda").manual_seed(9)
    for (M, C) in [(200704, 64), (50176, 256), (3136, 2048), (77, 24), (12544, 1024)]:
        x = torch.randn(M, C, device="cuda", generator=g).bfloat16()
        dy = torch.randn(M, C, device="cuda", generator=g).bfloat16()
        mean, invstd = x.float().mean(0), torch.rsqrt(x.float().var(0, unbiased=False) + 1e-5)
        gamma = torch.rand(C, device="cuda", generator=g) + 0.5
        outs = []
        for fused in (True, False, True):
            BN._FUSE_STAGE2 = fused
            try:
                dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
                dx, _ = BN.backward(x, dy, None, mean, invstd, gamma, dg, db, mask_coeffs=(gamma * invstd, -mean * gamma * invstd))
            finally:
                BN._FUSE_STAGE2 = False
            outs.append((dg, db, dx))
        for a, b in zip(outs[0], outs[1]):
            # the two forms add the partial rows in a different order: fp32 sums agree to ~1e-6, dx (bf16) to one rounding step
            tol = 2.0 ** -7 if a.dtype == torch.bfloat16 else 1e-4
            assert torch.allclose(a.float(), b.float(), rtol=tol, atol=tol * float(b.float().abs().max()))
        for a, b in zip(outs[0], outs[2]):
            assert torch.equal(a, b)                        # the fused form is deterministic
    cnt, _ = BN.ticket_ws(x.device)
    assert int(cnt.abs().sum()) == 0


def test_forward_and_dgrad_with_more_than_2_31_operand_elements():
    """64-bit operand offsets (the BIG instantiation): a 4.3 GB bf16 activation tensor. Every image of the big batch must equal, bit for
    bit, what the 32-bit kernel computes on that image alone (same tile arithmetic, same reduction order)."""
    from deeplearning4j_amd.ops import conv_igemm as IG
    free, _ = torch.cuda.mem_get_info()
    if free < 24 << 30:
        pytest.skip("needs ~20 GB of device memory")
    N, C, H, K = 64, 64, 725, 64
    assert N * C * H * H > 2 ** 31
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.empty((N, C, H, H), device="cuda", dtype=torch.bfloat16, memory_format=torch.channels_last)
    for n in range(N):                                   # filled per image: no fp32 temporary of the whole tensor
        x[n] = torch.randn(C, H, H, device="cuda", generator=g).bfloat16()
    w = (torch.randn(K, C, 1, 1, device="cuda", generator=g) / 8).bfloat16().contiguous(memory_format=torch.channels_last)
    assert IG.supported(x, w, (1, 1), (0, 0), (1, 1))
    y, st = IG.forward(x, w, None, (1, 1), (0, 0), (1, 1), want_stats=True)
    torch.cuda.synchronize()
    for n in (0, 31, 63):
        xs = x[n:n + 1].contiguous(memory_format=torch.channels_last)
        ys, _ = IG.forward(xs, w, None, (1, 1), (0, 0), (1, 1))
        assert torch.equal(y[n:n + 1], ys), n
    s = st.partial.view(st.gy, 2, K).double().sum(0)
    ref = torch.zeros(K, device="cuda", dtype=torch.float64)
    for n in range(N):
        ref += y[n].double().sum((1, 2))
    assert torch.allclose(s[0], ref, rtol=1e-4, atol=1e-3 * ref.abs().max().item())
    # data gradient: dz is the big operand now
    w_dgrad = torch.flip(w, dims=[2, 3]).permute(1, 2, 3, 0).contiguous()
    assert IG.dgrad_supported(w, (1, 1), (0, 0), y)
    dx = IG.dgrad(y, x.shape, w_dgrad, (1, 1), (0, 0), (1, 1))
    for n in (0, 63):
        d1 = IG.dgrad(y[n:n + 1].contiguous(memory_format=torch.channels_last), (1, C, H, H), w_dgrad, (1, 1), (0, 0), (1, 1))
        assert torch.equal(dx[n:n + 1], d1), n


@pytest.mark.parametrize("R,C,K,gy", [(1, 256, 64, 37), (3, 64, 64, 200), (1, 512, 2048, 9)])
def test_wgrad_launch_also_finishes_a_batchnorm_backward_reduction(R, C, K, gy):
    """dl4j_igemm_wgrad_tail: extra workgroups of the weight-gradient launch sum [gy][2][Cb] partial rows into sums[2 Cb] and the
    dbeta / dgamma views (the second stage of the BatchNorm backward that the preceding data-gradient launch started); dW itself must
    be what the plain launch produces."""
    from deeplearning4j_amd.ops import conv_igemm as IG
    g = torch.Generator(device="cuda").manual_seed(R * 1000 + C + K)
    N, H = 4, 14
    x = torch.randn(N, C, H, H, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    p = R // 2
    dz = torch.randn(N, K, H, H, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    dw_plain = IG.wgrad(dz, x, (K, C, R, R), (1, 1), (p, p), (1, 1))
    Cb = C                                            # the BatchNorm in front of this convolution has C channels
    part = torch.randn(gy, 2, Cb, device="cuda", generator=g)
    y2 = torch.empty(8, Cb, device="cuda", dtype=torch.bfloat16)
    db, dg = torch.full((Cb,), 7.0, device="cuda"), torch.full((Cb,), 7.0, device="cuda")
    sink = IG.BnBackwardSink(y2, None, None, dbeta_out=db, dgamma_out=dg)
    sink.partial, sink.gy = part.reshape(-1), gy
    assert sink.tail_ok()
    dw_tail = IG.wgrad(dz, x, (K, C, R, R), (1, 1), (p, p), (1, 1), tail=sink)
    assert sink.sums is not None and not sink.tail_ok()
    want = part.double().sum(0)
    assert torch.allclose(sink.sums.double(), want.reshape(-1), rtol=1e-5, atol=1e-4)
    assert torch.allclose(db.double(), want[0], rtol=1e-5, atol=1e-4) and torch.allclose(dg.double(), want[1], rtol=1e-5, atol=1e-4)
    sc = dw_plain.abs().max().item()
    assert (dw_tail - dw_plain).abs().max().item() <= 2e-3 * sc       # split-K atomics: order of the fp32 adds differs between runs


@pytest.mark.parametrize("M,nIn,nOut", [(4096, 768, 768), (1000, 256, 64), (520, 64, 136)])
@pytest.mark.parametrize("of_x", [False, True])
def test_wgrad_launch_also_sums_the_bias_gradient(M, nIn, nOut, of_x):
    """dl4j_igemm_wgrad_db: the dense-layer (1x1) weight-gradient launch accumulates the column sums of the gradient operand into db from
    the operand tiles it loads anyway - of the dy slot (mode 1) or of the x slot (mode 2: ops/linear.py passes the operands swapped to
    get the [nIn, nOut] image of dW). db is accumulated into (atomics), dW must be what the plain launch produces."""
    from deeplearning4j_amd.ops import conv_igemm as IG
    g = torch.Generator(device="cuda").manual_seed(M + nIn + nOut)
    x = torch.randn(M, nIn, device="cuda", generator=g).bfloat16()
    dz = torch.randn(M, nOut, device="cuda", generator=g).bfloat16()
    a, b = (x, dz) if of_x else (dz, x)              # (dy slot, x slot)
    ka, kb = a.shape[1], b.shape[1]
    plain = IG.wgrad(a.view(M, ka, 1, 1), b.view(M, kb, 1, 1), (ka, kb, 1, 1), (1, 1), (0, 0), (1, 1))
    db = torch.full((nOut,), 3.0, device="cuda")
    fused = IG.wgrad(a.view(M, ka, 1, 1), b.view(M, kb, 1, 1), (ka, kb, 1, 1), (1, 1), (0, 0), (1, 1), db_out=db, db_of_x=of_x)
    want = 3.0 + dz.double().sum(0)
    assert torch.allclose(db.double(), want, rtol=1e-4, atol=2e-3 * float(dz.float().abs().sum(0).max()) / 100 + 1e-3), (db.double() - want).abs().max()
    sc = plain.abs().max().item()
    assert (fused - plain).abs().max().item() <= 2e-3 * sc
