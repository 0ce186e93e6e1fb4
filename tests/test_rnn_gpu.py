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


@pytest.mark.parametrize("T,B,nIn,n", [(6, 64, 96, 128), (9, 100, 64, 256), (5, 128, 128, 512)])
@pytest.mark.parametrize("bf", [torch.bfloat16, torch.float32])
def test_gru_fused_step_kernels_match_torch_gru_forward_and_backward(T, B, nIn, n, bf):
    """gru_seq.hip (one launch per step per direction, bf16 and fp32 instantiations) against torch.nn.GRU in fp32 with the same weights:
    libnd4j's gate order [r | u | c] is ATen's (r, z, n) and the candidate uses r * (h . Wh_c), so the two are the same function."""
    from deeplearning4j_amd.nd4j.ops.custom import exec_op
    from deeplearning4j_amd.ops import lstm as R
    g = torch.Generator(device="cuda").manual_seed(11)
    f32 = bf == torch.float32
    x = torch.randn(T, B, nIn, device="cuda", generator=g).to(bf)
    h0 = (torch.randn(B, n, device="cuda", generator=g) * 0.5).to(bf)
    Wx = (torch.randn(nIn, 3 * n, device="cuda", generator=g) / nIn ** 0.5).to(bf)
    Wh = (torch.randn(n, 3 * n, device="cuda", generator=g) / n ** 0.5).to(bf)
    b = (torch.randn(3 * n, device="cuda", generator=g) * 0.1).to(bf)
    m = torch.nn.GRU(nIn, n).cuda().float()
    with torch.no_grad():
        m.weight_ih_l0.copy_(Wx.float().t()); m.weight_hh_l0.copy_(Wh.float().t()); m.bias_ih_l0.copy_(b.float()); m.bias_hh_l0.zero_()
    xr, hr = x.float().requires_grad_(True), h0.float().requires_grad_(True)
    ref = m(xr, hr.unsqueeze(0))[0]
    dH = torch.randn(T, B, n, device="cuda", generator=g)
    rgx, rgh, rgwi, rgwh, rgb = torch.autograd.grad(ref, [xr, hr, m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0], dH)
    ins = [v.detach().clone().requires_grad_(True) for v in (x, h0, Wx, Wh, b)]
    assert R.gru_fused_ok(torch.empty(T, B, 3 * n, device="cuda", dtype=bf), n)
    out = exec_op("gru", ins)[0]
    assert type(out.grad_fn).__name__.startswith("_GruSeqFused")
    scale = float(ref.abs().max())
    assert (out.float() - ref).abs().max().item() < (2e-4 if f32 else 3e-2) * max(1.0, scale)
    gg = torch.autograd.grad(out, ins, dH.to(bf))
    for name, a_, r_ in zip(("dx", "dh0", "dWx", "dWh", "db"), gg, (rgx, rgh, rgwi.t(), rgwh.t(), rgb)):
        err = (a_.float() - r_).abs().max().item()
        cos = float((a_.float() * r_).sum() / (a_.float().norm() * r_.norm() + 1e-12))
        assert cos > (0.999999 if f32 else 0.995) and err < (1e-3 if f32 else 6e-2) * max(1.0, r_.abs().max().item()), (name, err, cos)
    # inference form: no saved activations
    with torch.no_grad():
        out2 = exec_op("gru", [x, h0, Wx, Wh, b])[0]
    assert torch.equal(out2, out.detach())


def _seq_case(T, mb, n, peep, masked, seed):
    g = torch.Generator().manual_seed(seed)
    xw = (torch.randn(T, mb, 4 * n, generator=g) * 0.5).bfloat16().cuda()
    rw4 = (torch.randn(4 * n, n, generator=g) / n ** 0.5).bfloat16().cuda()
    RW4 = rw4.t()                                   # DL4J 'f'-order [n, 4n] whose memory image is the [4n, n] operand
    h0 = (torch.randn(mb, n, generator=g) * 0.3).bfloat16().cuda()
    c0 = (torch.randn(mb, n, generator=g) * 0.3).bfloat16().cuda()
    pp = [(torch.randn(n, generator=g) * 0.2).cuda() for _ in range(3)] if peep else None
    mask = (torch.rand(mb, T, generator=g) > 0.2).float().cuda() if masked else None
    return xw, RW4, h0, c0, pp, mask


@pytest.mark.gpu
@pytest.mark.parametrize("T,mb,n,peep,masked,reverse", [(7, 64, 512, False, False, False), (5, 33, 128, True, True, False),
                                                        (6, 100, 256, True, False, True), (1, 8, 64, False, True, False),
                                                        (50, 64, 512, False, False, False)])
@pytest.mark.parametrize("mode", ["steps", "grid"])
def test_lstm_persistent_forward_matches_per_step_kernels(monkeypatch, mode, T, mb, n, peep, masked, reverse):
    """One-launch persistent sequence kernel (lstm_persist.hip) against the per-step GEMM + cell kernels, and both against fp32 torch math."""
    from deeplearning4j_amd.ops import lstm as R
    xw, RW4, h0, c0, pp, mask = _seq_case(T, mb, n, peep, masked, 11)
    monkeypatch.setattr(R, "_PERSIST_MODE", mode)
    monkeypatch.setattr(R, "_PERSIST", True)
    monkeypatch.setattr(R, "_PERSIST_CHECK", True)
    assert R.persist_ok(xw, RW4, mb, n)
    Hp, stp = R.seq_forward(xw, RW4, h0, c0, pp, mask, True, reverse)
    monkeypatch.setattr(R, "_PERSIST", False)
    Hs, sts = R.seq_forward(xw, RW4, h0, c0, pp, mask, True, reverse)
    # fp32 reference of the recurrence
    W = RW4.float()
    h, c = h0.float(), c0.float()
    Href = torch.zeros(T, mb, n, device="cuda")
    for s in range(T):
        t = T - 1 - s if reverse else s
        z = xw[t].float() + h @ W
        za, zf, zo, zg = z[:, :n], z[:, n:2 * n], z[:, 2 * n:3 * n], z[:, 3 * n:]
        if pp:
            zf, zg = zf + c * pp[0], zg + c * pp[2]
        a, f, g_ = torch.tanh(za), torch.sigmoid(zf), torch.sigmoid(zg)
        cn = f * c + g_ * a
        if pp:
            zo = zo + cn * pp[1]
        hn = torch.sigmoid(zo) * torch.tanh(cn)
        if mask is not None:
            m = mask[:, t].unsqueeze(1)
            hn, cn = hn * m, cn * m
        # the kernels carry h / c between steps in bf16
        h, c = hn.bfloat16().float(), cn.bfloat16().float()
        Href[t] = hn
    tol = dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(Hp.float(), Href, **tol)
    torch.testing.assert_close(Hs.float(), Href, **tol)
    torch.testing.assert_close(Hp.float(), Hs.float(), **tol)
    torch.testing.assert_close(stp["Cs"].float(), sts["Cs"].float(), **tol)
    torch.testing.assert_close(stp["SV"].float(), sts["SV"].float(), **tol)
    assert float((Hp.float() - Href).abs().mean()) < 3e-3


@pytest.mark.gpu
@pytest.mark.parametrize("T,mb,n,peep,masked,reverse", [(7, 64, 512, False, False, False), (5, 33, 128, True, True, False),
                                                        (6, 100, 256, True, False, True), (1, 8, 64, False, True, False),
                                                        (50, 64, 512, True, True, False)])
@pytest.mark.parametrize("mode", ["steps", "grid"])
def test_lstm_persistent_backward_matches_per_step_kernels_and_autograd(monkeypatch, mode, T, mb, n, peep, masked, reverse):
    """Persistent backward kernel: gate gradients + peephole gradients against the per-step kernels, and against torch autograd of
    the fp32 recurrence (same saved forward in both kernel paths)."""
    from deeplearning4j_amd.ops import lstm as R
    xw, RW4, h0, c0, pp, mask = _seq_case(T, mb, n, peep, masked, 17)
    g = torch.Generator().manual_seed(5)
    E = (torch.randn(T, mb, n, generator=g) * 0.5).bfloat16().cuda()
    monkeypatch.setattr(R, "_PERSIST_CHECK", True)
    monkeypatch.setattr(R, "_PERSIST_MODE", mode)
    monkeypatch.setattr(R, "_PERSIST", True)
    H, st = R.seq_forward(xw, RW4, h0, c0, pp, mask, True, reverse)
    dGp, Hprev_p, dpp = R.seq_backward(E, RW4, st, reverse)
    monkeypatch.setattr(R, "_PERSIST", False)
    dGs, Hprev_s, dps = R.seq_backward(E, RW4, st, reverse)
    assert torch.equal(Hprev_p, Hprev_s)
    # autograd reference in fp32 on the same inputs
    xw32 = xw.float().requires_grad_(True)
    W = RW4.float()
    ppr = [q.clone().requires_grad_(True) for q in pp] if pp else None
    h, c = h0.float(), c0.float()
    Hs_ = [None] * T
    for s in range(T):
        t = T - 1 - s if reverse else s
        z = xw32[t] + h @ W
        za, zf, zo, zg = z[:, :n], z[:, n:2 * n], z[:, 2 * n:3 * n], z[:, 3 * n:]
        if ppr:
            zf, zg = zf + c * ppr[0], zg + c * ppr[2]
        a, f, g_ = torch.tanh(za), torch.sigmoid(zf), torch.sigmoid(zg)
        cn = f * c + g_ * a
        if ppr:
            zo = zo + cn * ppr[1]
        hn = torch.sigmoid(zo) * torch.tanh(cn)
        if mask is not None:
            m = mask[:, t].unsqueeze(1)
            hn, cn = hn * m, cn * m
        h, c = hn, cn
        Hs_[t] = hn
    (torch.stack(Hs_) * E.float()).sum().backward()
    ref = xw32.grad
    scale = float(ref.abs().mean())
    for dG in (dGp, dGs):
        err = (dG.float() - ref).abs()
        assert float(err.mean()) < 0.03 * scale + 1e-4, (float(err.mean()), scale)
        assert float(err.max()) < 0.25 * float(ref.abs().max()) + 1e-2
    assert float((dGp.float() - dGs.float()).abs().mean()) < 0.02 * scale + 1e-4
    # the persistent kernel keeps dc / dh in fp32 between steps: it must be at least as close to autograd as the bf16 per-step path
    assert float((dGp.float() - ref).abs().mean()) <= float((dGs.float() - ref).abs().mean()) * 1.05 + 1e-5
    if pp:
        for k, q in enumerate((0, 1, 2)):
            refp = ppr[q].grad
            torch.testing.assert_close(dpp[k], refp, rtol=0.08, atol=0.05 * float(refp.abs().max()) + 1e-3)
            torch.testing.assert_close(dps[k], refp, rtol=0.08, atol=0.05 * float(refp.abs().max()) + 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("T,mb,n,peep,masked,reverse", [(6, 64, 512, False, False, False), (5, 33, 128, True, True, False),
                                                        (4, 100, 256, True, False, True), (1, 8, 64, False, True, False)])
def test_lstm_fp32_step_kernels_match_autograd(monkeypatch, T, mb, n, peep, masked, reverse):
    """fp32 fused step kernels (fp32 MFMA) against torch autograd of the same recurrence, and against the per-step GEMM + cell path."""
    from deeplearning4j_amd.ops import lstm as R
    g = torch.Generator().manual_seed(23)
    xw = (torch.randn(T, mb, 4 * n, generator=g) * 0.5).cuda()
    RW4 = (torch.randn(4 * n, n, generator=g) / n ** 0.5).cuda().t()
    h0, c0 = (torch.randn(mb, n, generator=g) * 0.3).cuda(), (torch.randn(mb, n, generator=g) * 0.3).cuda()
    pp = [(torch.randn(n, generator=g) * 0.2).cuda() for _ in range(3)] if peep else None
    mask = (torch.rand(mb, T, generator=g) > 0.2).float().cuda() if masked else None
    E = (torch.randn(T, mb, n, generator=g) * 0.5).cuda()
    monkeypatch.setattr(R, "_PERSIST", True)
    assert R.steps_f32_ok(xw, RW4, mb, n)
    H, st = R.seq_forward(xw, RW4, h0, c0, pp, mask, True, reverse)
    dG, Hprev, dp = R.seq_backward(E, RW4, st, reverse)
    monkeypatch.setattr(R, "_PERSIST", False)
    Hs, sts = R.seq_forward(xw, RW4, h0, c0, pp, mask, True, reverse)
    dGs, _, dps = R.seq_backward(E, RW4, sts, reverse)
    xr = xw.clone().requires_grad_(True)
    ppr = [q.clone().requires_grad_(True) for q in pp] if pp else None
    W = RW4
    h, c = h0, c0
    Hl = [None] * T
    for s in range(T):
        t = T - 1 - s if reverse else s
        z = xr[t] + h @ W
        za, zf, zo, zg = z[:, :n], z[:, n:2 * n], z[:, 2 * n:3 * n], z[:, 3 * n:]
        if ppr:
            zf, zg = zf + c * ppr[0], zg + c * ppr[2]
        a, f, g_ = torch.tanh(za), torch.sigmoid(zf), torch.sigmoid(zg)
        cn = f * c + g_ * a
        if ppr:
            zo = zo + cn * ppr[1]
        hn = torch.sigmoid(zo) * torch.tanh(cn)
        if mask is not None:
            m = mask[:, t].unsqueeze(1)
            hn, cn = hn * m, cn * m
        h, c = hn, cn
        Hl[t] = hn
    Href = torch.stack(Hl)
    (Href * E).sum().backward()
    torch.testing.assert_close(H, Href.detach(), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(Hs, Href.detach(), rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(dG, xr.grad, rtol=2e-3, atol=5e-4)
    torch.testing.assert_close(dGs, xr.grad, rtol=2e-3, atol=5e-4)
    if pp:
        for k in range(3):
            torch.testing.assert_close(dp[k], ppr[k].grad, rtol=2e-3, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,fmt,dir_mode,peep", [(torch.float32, 0, 0, False), (torch.float32, 1, 1, True), (torch.float32, 2, 3, True),
                                                    (torch.bfloat16, 1, 0, False), (torch.float32, 0, 2, False), (torch.bfloat16, 0, 4, True)])
def test_lstm_layer_op_fused_route_matches_generic(dtype, fmt, dir_mode, peep):
    """lstmLayer custom op: the fused-kernel platform route against the generic implementation (forward values and all gradients)."""
    from deeplearning4j_amd.nd4j.environment import Environment
    from deeplearning4j_amd.nd4j.ops.custom import OpExecutioner, exec_op
    T, B, nIn, n = 9, 16, 24, 64
    g = torch.Generator().manual_seed(31)
    bid = dir_mode >= 2
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.3)
    x = mk(T, B, nIn)
    x = {0: x, 1: x.permute(1, 0, 2), 2: x.permute(1, 2, 0), 3: x}[fmt].contiguous()
    Wx, Wr, bias = mk(*((2,) if bid else ()), nIn, 4 * n), mk(*((2,) if bid else ()), n, 4 * n), mk(*((2,) if bid else ()), 4 * n)
    hI, cI = mk(*((2,) if bid else ()), B, n), mk(*((2,) if bid else ()), B, n)
    Wp = mk(*((2,) if bid else ()), 3 * n) if peep else None
    bargs = [True, False, True, True, peep, True, True, True]          # full sequence, last h and last c returned; all of them get gradients
    iargs = [fmt, dir_mode, 2, 0, 0]

    def run(helpers):
        leaves = [v.to(dtype).cuda().requires_grad_(True) for v in [x, Wx, Wr, bias, hI, cI] + ([Wp] if peep else [])]
        Environment.getInstance().allowHelpers(helpers)
        try:
            OpExecutioner.resetHelperStatistics()
            outs = exec_op("lstmLayer", leaves, iArgs=iargs, tArgs=[0.0], bArgs=bargs)
            hit = OpExecutioner.helperStatistics()["hit"].get("lstmLayer", 0)
        finally:
            Environment.getInstance().allowHelpers(True)
        w = [torch.randn(o.shape, generator=torch.Generator().manual_seed(7 + k)).to(o.device) for k, o in enumerate(outs)]
        loss = sum((o.float() * wk).sum() for o, wk in zip(outs, w))
        req = [v for v in leaves if v.requires_grad]
        grads = torch.autograd.grad(loss, req)
        return [o.detach().float() for o in outs], [gr.float() for gr in grads], hit

    o_h, g_h, hit_h = run(True)
    o_g, g_g, hit_g = run(False)
    assert hit_h == 1 and hit_g == 0
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float32 else dict(rtol=6e-2, atol=6e-2)
    for a, b_ in zip(o_h, o_g):
        assert a.shape == b_.shape
        torch.testing.assert_close(a, b_, **tol)
    for a, b_ in zip(g_h, g_g):
        assert a.shape == b_.shape
        if dtype == torch.float32:
            torch.testing.assert_close(a, b_, **tol)
        else:
            assert float((a - b_).abs().mean()) < 0.05 * float(b_.abs().mean()) + 1e-3


@pytest.mark.parametrize("capture", [False, True])
def test_bidirectional_two_stream_overlap_matches_single_stream(monkeypatch, capture):
    """The two directions of GravesBidirectionalLSTM / Bidirectional run on two HIP streams (nn/layers/recurrent.py _side_stream): same
    numbers as the in-line order, eager and inside a captured training step."""
    x = torch.randn(8, 12, 9, device="cuda")
    y = torch.nn.functional.one_hot(torch.randint(0, 5, (8, 9), device="cuda"), 5).permute(0, 2, 1).float()
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("DL4J_AMD_BIDIR_STREAMS", flag)
        net = _net("bidir", 12, 64, "BFLOAT16", "cuda")
        if capture:
            net.enableGraphCapture(True)
        for _ in range(4):
            net.fit(x, y)
        torch.cuda.synchronize()
        out[flag] = (net.params().clone(), float(net.score()))
    assert torch.isfinite(out["1"][0]).all()
    assert torch.equal(out["0"][0], out["1"][0]), (out["0"][0] - out["1"][0]).abs().max().item()
    assert abs(out["0"][1] - out["1"][1]) < 1e-5 * abs(out["0"][1])   # the score reduction is not bit-reproducible


def _stack_net(kind, dtype, n=64, nIn=12, T=24, tbptt=None, seed=5):
    from deeplearning4j_amd.nd4j.learning import Sgd
    from deeplearning4j_amd.nn.conf.base import InputType, BackpropType
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import LSTM, GravesLSTM, RnnOutputLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    b = NeuralNetConfiguration.Builder().seed(seed).updater(Sgd(0.05)).weightInit("XAVIER")
    if dtype:
        b = b.computeDataType(dtype)
    Lc = {"lstm": LSTM, "graves": GravesLSTM}[kind]
    lb = b.list().layer(Lc(nIn=nIn, nOut=n, activation="TANH")).layer(Lc(nIn=n, nOut=2 * n, activation="TANH")).layer(Lc(nIn=2 * n, nOut=n, activation="TANH"))
    lb = lb.layer(RnnOutputLayer(nIn=n, nOut=5, activation="SOFTMAX", lossFunction="MCXENT"))
    if tbptt:
        lb = lb.backpropType(BackpropType.TruncatedBPTT).tBPTTForwardLength(tbptt).tBPTTBackwardLength(tbptt)
    return MultiLayerNetwork(lb.setInputType(InputType.recurrent(nIn, T)).build(), device="cuda").init()


@pytest.mark.parametrize("kind", ["lstm", "graves"])
@pytest.mark.parametrize("dtype", [None, "BFLOAT16"])
def test_lstm_stack_pipeline_matches_layer_by_layer(monkeypatch, kind, dtype):
    """Three stacked LSTM layers run as a layer x chunk wavefront over several HIP streams (nn/layers/lstm_stack.py): score, activations and
    every gradient equal the layer-by-layer schedule (the chunked projection GEMMs may round differently from the one big GEMM)."""
    from deeplearning4j_amd.nn.layers import lstm_stack
    calls = {"f": 0, "b": 0}
    rf, rb = lstm_stack.forward, lstm_stack.backward
    monkeypatch.setattr(lstm_stack, "forward", lambda *a, **k: (calls.__setitem__("f", calls["f"] + 1), rf(*a, **k))[1])
    monkeypatch.setattr(lstm_stack, "backward", lambda *a, **k: (calls.__setitem__("b", calls["b"] + 1), rb(*a, **k))[1])
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 12, 24, generator=g).cuda()
    y = torch.nn.functional.one_hot(torch.randint(0, 5, (8, 24), generator=g), 5).permute(0, 2, 1).float().cuda()
    res = {}
    for on in (False, True):
        monkeypatch.setattr(lstm_stack, "ENABLED", on)
        net = _stack_net(kind, dtype)
        out = net.output(x).float()
        net.computeGradientAndScore(x, y)
        torch.cuda.synchronize()
        res[on] = (out, float(net._score), net._grads.float().clone())
    assert calls["f"] == 2 and calls["b"] == 1, calls
    tol = 2e-2 if dtype else 1e-4
    assert (res[True][0] - res[False][0]).abs().max().item() < tol
    assert abs(res[True][1] - res[False][1]) < tol * max(1.0, abs(res[False][1]))
    ga, gb = res[False][2].double(), res[True][2].double()
    assert float((ga * gb).sum() / (ga.norm() * gb.norm())) > (0.9995 if dtype else 0.999999)
    assert (ga - gb).abs().max().item() < tol * max(1.0, ga.abs().max().item())


def _equal_stack_net(kind, dtype, n=128, nIn=12, T=24, seed=5, tbptt=None):
    from deeplearning4j_amd.nd4j.learning import Sgd
    from deeplearning4j_amd.nn.conf.base import InputType, BackpropType
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import LSTM, GravesLSTM, RnnOutputLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    b = NeuralNetConfiguration.Builder().seed(seed).updater(Sgd(0.05)).weightInit("XAVIER")
    if dtype:
        b = b.computeDataType(dtype)
    Lc = {"lstm": LSTM, "graves": GravesLSTM}[kind]
    lb = b.list().layer(Lc(nIn=nIn, nOut=n, activation="TANH")).layer(Lc(nIn=n, nOut=n, activation="TANH")).layer(Lc(nIn=n, nOut=n, activation="TANH"))
    lb = lb.layer(RnnOutputLayer(nIn=n, nOut=5, activation="SOFTMAX", lossFunction="MCXENT"))
    if tbptt:
        lb = lb.backpropType(BackpropType.TruncatedBPTT).tBPTTForwardLength(tbptt).tBPTTBackwardLength(tbptt)
    return MultiLayerNetwork(lb.setInputType(InputType.recurrent(nIn, T)).build(), device="cuda").init()


@pytest.mark.parametrize("kind", ["lstm", "graves"])
@pytest.mark.parametrize("mb,T", [(8, 24), (40, 11)])
def test_lstm_stack_diagonal_launches_match_chunk_pipeline_and_fp32(monkeypatch, kind, mb, T):
    """bf16 stacks of equal width run one launch per anti-diagonal of the layer x time wavefront, the upper layers' input projection
    folded into the step as a second K segment (lstm_diag_fwd_kernel / lstm_diag_bwd_kernel). Against the chunked multi-stream pipeline
    (same bf16 math, projections rounded to bf16 once more) and against the fp32 layer-by-layer schedule."""
    from deeplearning4j_amd.nn.layers import lstm_stack
    used = {"n": 0}
    inner = lstm_stack._diag_ok
    monkeypatch.setattr(lstm_stack, "_diag_ok", lambda *a, **k: (lambda r: (used.__setitem__("n", used["n"] + int(r)), r)[1])(inner(*a, **k)))
    g = torch.Generator().manual_seed(17)
    x = torch.randn(mb, 12, T, generator=g).cuda()
    y = torch.nn.functional.one_hot(torch.randint(0, 5, (mb, T), generator=g), 5).permute(0, 2, 1).float().cuda()
    res = {}
    for name, dtype, diag in (("diag", "BFLOAT16", True), ("chunks", "BFLOAT16", False), ("fp32", None, False)):
        monkeypatch.setattr(lstm_stack, "DIAG", diag)
        monkeypatch.setattr(lstm_stack, "ENABLED", name != "fp32")
        net = _equal_stack_net(kind, dtype, T=T)
        out = net.output(x).float()
        net.computeGradientAndScore(x, y)
        torch.cuda.synchronize()
        res[name] = (out, float(net._score), net._grads.float().clone())
    assert used["n"] == 3, used                      # inference forward, training forward, backward
    for ref, tol, cos in (("chunks", 1e-2, 0.9995), ("fp32", 3e-2, 0.999)):
        assert (res["diag"][0] - res[ref][0]).abs().max().item() < tol
        assert abs(res["diag"][1] - res[ref][1]) < tol * max(1.0, abs(res[ref][1]))
        ga, gb = res[ref][2].double(), res["diag"][2].double()
        assert float((ga * gb).sum() / (ga.norm() * gb.norm())) > cos
        assert (ga - gb).abs().max().item() < tol * max(1.0, ga.abs().max().item())


@pytest.mark.parametrize("kind", ["lstm", "graves"])
def test_lstm_stack_diagonal_launches_fp32_match_layer_by_layer(monkeypatch, kind):
    """The fp32 instantiation of the diagonal kernels (MFMA 32x32x2 / 16x16x4) against the layer-by-layer schedule."""
    from deeplearning4j_amd.nn.layers import lstm_stack
    used = {"n": 0}
    inner = lstm_stack._diag_ok
    monkeypatch.setattr(lstm_stack, "_diag_ok", lambda *a, **k: (lambda r: (used.__setitem__("n", used["n"] + int(r)), r)[1])(inner(*a, **k)))
    g = torch.Generator().manual_seed(19)
    x = torch.randn(24, 12, 13, generator=g).cuda()
    y = torch.nn.functional.one_hot(torch.randint(0, 5, (24, 13), generator=g), 5).permute(0, 2, 1).float().cuda()
    res = {}
    for on in (False, True):
        monkeypatch.setattr(lstm_stack, "ENABLED", on)
        net = _equal_stack_net(kind, None, T=13)
        out = net.output(x).float()
        net.computeGradientAndScore(x, y)
        torch.cuda.synchronize()
        res[on] = (out, float(net._score), net._grads.float().clone())
    assert used["n"] == 3, used
    assert (res[True][0] - res[False][0]).abs().max().item() < 1e-4
    assert abs(res[True][1] - res[False][1]) < 1e-4 * max(1.0, abs(res[False][1]))
    ga, gb = res[False][2].double(), res[True][2].double()
    assert float((ga * gb).sum() / (ga.norm() * gb.norm())) > 0.999999
    assert (ga - gb).abs().max().item() < 1e-4 * max(1.0, ga.abs().max().item())


def test_lstm_stack_diagonal_tbptt_fit_under_graph_capture(monkeypatch):
    """Truncated BPTT with the diagonal launches captured in a hipGraph: state carried between segments and parameters after three
    fits equal the eager run bit for bit."""
    from deeplearning4j_amd.nn.layers import lstm_stack
    g = torch.Generator().manual_seed(8)
    x = torch.randn(8, 12, 48, generator=g).cuda()
    y = torch.nn.functional.one_hot(torch.randint(0, 5, (8, 48), generator=g), 5).permute(0, 2, 1).float().cuda()
    res = {}
    for capture in (False, True):
        net = _equal_stack_net("graves", "BFLOAT16", T=48, tbptt=16)
        if capture:
            net.enableGraphCapture(True)
        for _ in range(3):
            net.fit(x, y)
        torch.cuda.synchronize()
        res[capture] = (net.params().clone(), float(net.score()))
    assert torch.isfinite(res[True][0]).all()
    assert torch.equal(res[True][0], res[False][0])


@pytest.mark.parametrize("capture", [False, True])
def test_lstm_stack_pipeline_tbptt_fit(monkeypatch, capture):
    """Truncated BPTT over the pipelined stack: the state carried between segments and the parameters after two fits match the
    layer-by-layer schedule, eager and with the training step captured in a hipGraph."""
    from deeplearning4j_amd.nn.layers import lstm_stack
    g = torch.Generator().manual_seed(8)
    x = torch.randn(8, 12, 48, generator=g).cuda()
    y = torch.nn.functional.one_hot(torch.randint(0, 5, (8, 48), generator=g), 5).permute(0, 2, 1).float().cuda()
    res = {}
    for on in (False, True):
        monkeypatch.setattr(lstm_stack, "ENABLED", on)
        net = _stack_net("graves", None, T=48, tbptt=16)
        if capture:
            net.enableGraphCapture(True)
        for _ in range(3):
            net.fit(x, y)
        torch.cuda.synchronize()
        res[on] = (net.params().clone(), float(net.score()))
    assert torch.isfinite(res[True][0]).all()
    assert (res[True][0] - res[False][0]).abs().max().item() < 1e-4
    assert abs(res[True][1] - res[False][1]) < 1e-4 * max(1.0, abs(res[False][1]))


@pytest.mark.parametrize("dtype", [None, "BFLOAT16"])
def test_graph_lstm_chain_pipeline_matches_vertex_by_vertex(monkeypatch, dtype):
    """ComputationGraph: a chain of LSTM vertices runs through the same multi-stream pipeline (_LstmStackNode)."""
    from deeplearning4j_amd.nn.layers import lstm_stack
    from tests.test_networks_cpu import _lstm_chain_graph
    calls = {"f": 0, "b": 0}
    rf, rb = lstm_stack.forward, lstm_stack.backward
    monkeypatch.setattr(lstm_stack, "forward", lambda *a, **k: (calls.__setitem__("f", calls["f"] + 1), rf(*a, **k))[1])
    monkeypatch.setattr(lstm_stack, "backward", lambda *a, **k: (calls.__setitem__("b", calls["b"] + 1), rb(*a, **k))[1])
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 6, 12, generator=g).cuda()
    y = torch.nn.functional.one_hot(torch.randint(0, 3, (8, 12), generator=g), 3).permute(0, 2, 1).float().cuda()
    res = {}
    for on in (False, True):
        monkeypatch.setattr(lstm_stack, "ENABLED", on)
        cg = _lstm_chain_graph("cuda", dtype, n=64)
        cg.computeGradientAndScore([x], [y])
        torch.cuda.synchronize()
        res[on] = (float(cg._score), cg._grads.float().clone())
    assert calls == {"f": 1, "b": 1}, calls
    tol = 2e-2 if dtype else 1e-4
    assert abs(res[True][0] - res[False][0]) < tol * max(1.0, abs(res[False][0]))
    ga, gb = res[False][1].double(), res[True][1].double()
    assert float((ga * gb).sum() / (ga.norm() * gb.norm())) > (0.9995 if dtype else 0.999999)
