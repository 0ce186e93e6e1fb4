"""fused_dense (GEMM + bias / activation / residual epilogues, activation-derivative + bias-gradient pass, direct gradient
accumulation through the mixed-precision Links), the [B, T, H, D]-layout attention kernels and the LinearFusion graph pass, on the GPU
against fp32 PyTorch math."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, w, b, r, act):
    from deeplearning4j_amd.ops.linear import fused_dense_reference
    return fused_dense_reference(x, w, b, r, act)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("act,res", [(None, False), ("gelu", False), ("precise_gelu", False), ("gelu_erf", False), ("tanh", True),
                                     (None, True), ("relu", False), ("swish", False)])
def test_fused_dense_forward_backward_matches_fp32(dtype, act, res):
    import deeplearning4j_amd.ops.linear as LIN
    g = torch.Generator(device="cuda").manual_seed(1)
    M, K, N = 200, 96, 136
    x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    w = (torch.randn(K, N, device="cuda", generator=g) / 8).to(dtype)
    b = torch.randn(N, device="cuda", generator=g).to(dtype)
    r = torch.randn(M, N, device="cuda", generator=g).to(dtype) if res else None
    ins = [t.detach().clone().requires_grad_(True) if t is not None else None for t in (x, w, b, r)]
    rin = [t.detach().float().clone().requires_grad_(True) if t is not None else None for t in (x, w, b, r)]
    out = LIN.fused_dense(*ins, act=act)
    ref = _ref(*rin, act)
    gy = torch.randn_like(ref)
    got = torch.autograd.grad(out, [t for t in ins if t is not None], gy.to(dtype))
    want = torch.autograd.grad(ref, [t for t in rin if t is not None], gy)
    tol = 3e-4 if dtype == torch.float32 else 6e-2
    assert (out.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    for a_, w_ in zip(got, want):
        assert (a_.float() - w_).abs().max().item() < tol * max(1.0, w_.abs().max().item())
    # inference form (epilogue activation where the GEMM knows it)
    with torch.no_grad():
        o2 = LIN.fused_dense(x, w, b, r, act=act)
    assert (o2.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


def test_links_accumulate_weight_and_bias_gradients_in_place():
    """A compute-precision leaf linked to fp32 master / gradient views: forward reads the transposed replica and the fp32 bias,
    backward leaves dW / db in the flat gradient views (autograd sees no gradient for them); a second use of the same weight adds."""
    import deeplearning4j_amd.ops.linear as LIN
    g = torch.Generator(device="cuda").manual_seed(2)
    M, K, N = 128, 64, 72
    flat = torch.randn(K * N + N, device="cuda", generator=g) / 8
    flat_g = torch.zeros_like(flat)
    shadow = torch.empty(2 * K * N + N + 8, device="cuda", dtype=torch.bfloat16)
    shadow[:K * N + N].copy_(flat)
    t_off = (K * N + N + 7) // 8 * 8
    LIN.TransposeTable([(0, t_off, K, N)], flat.device).run(flat, shadow)
    wt = shadow[t_off:t_off + K * N].view(N, K)
    assert torch.equal(wt, shadow[:K * N].view(K, N).t())
    w = shadow[:K * N].view(K, N).detach().requires_grad_(True)
    b = shadow[K * N:K * N + N].detach().requires_grad_(True)
    links = {w.data_ptr(): LIN.Link(flat[:K * N], flat_g[:K * N], wt, K * N, torch.bfloat16),
             b.data_ptr(): LIN.Link(flat[K * N:], flat_g[K * N:], None, N, torch.bfloat16)}
    x1 = torch.randn(M, K, device="cuda", generator=g).bfloat16().requires_grad_(True)
    x2 = torch.randn(M, K, device="cuda", generator=g).bfloat16().requires_grad_(True)
    LIN.set_links(links)
    try:
        y = LIN.fused_dense(x1, w, b, act="gelu") + LIN.fused_dense(x2, w, b)
        gy = torch.randn_like(y)
        dx1, dx2, dw, db = torch.autograd.grad(y, [x1, x2, w, b], gy, allow_unused=True)
    finally:
        LIN.set_links(None)
    assert dw is None and db is None
    wf, bf = flat[:K * N].view(K, N).bfloat16().float().requires_grad_(True), flat[K * N:].clone().requires_grad_(True)
    a1, a2 = x1.detach().float().requires_grad_(True), x2.detach().float().requires_grad_(True)
    z = a1 @ wf + bf
    ref = z * torch.sigmoid(1.702 * z) + a2 @ wf + bf
    r1, r2, rw, rb = torch.autograd.grad(ref, [a1, a2, wf, bf], gy.float())
    assert (flat_g[:K * N].view(K, N) - rw).abs().max().item() < 5e-2 * rw.abs().max().item()
    assert (flat_g[K * N:] - rb).abs().max().item() < 5e-2 * rb.abs().max().item()
    assert (dx1.float() - r1).abs().max().item() < 5e-2 * r1.abs().max().item()
    assert (dx2.float() - r2).abs().max().item() < 5e-2 * r2.abs().max().item()


@pytest.mark.parametrize("masked", [False, True])
def test_sdpa_bthd_layout_matches_contiguous(masked):
    """q / k / v as per-head views of [B*T, H*D] linear outputs: same results as the contiguous [B, H, T, D] path, no copies, and the
    output / gradients come back in the layout whose [B*T, H*D] reshape is a view."""
    from deeplearning4j_amd.ops import attention as ATT
    g = torch.Generator(device="cuda").manual_seed(3)
    B, H, T, D = 3, 4, 160, 64
    base = [torch.randn(B * T, H * D, device="cuda", generator=g).bfloat16() for _ in range(3)]
    mask = None
    if masked:
        mask = torch.ones(B, T, device="cuda")
        mask[1, 100:] = 0
        mask[2, 7:] = 0
    def run(contig):
        leaves = [t.detach().clone().requires_grad_(True) for t in base]
        qkv = [t.view(B, T, H, D).permute(0, 2, 1, 3) for t in leaves]
        if contig:
            qkv = [t.contiguous() for t in qkv]
        o = ATT.sdpa(*qkv, mask=mask, scale=1.0 / math.sqrt(D))
        o2 = o.permute(0, 2, 1, 3).reshape(B * T, H * D)
        if not contig:
            assert o2.data_ptr() == o.data_ptr()            # the reshape is a view of the kernel's output
        gy = torch.randn(B * T, H * D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)).bfloat16()
        return (o2,) + torch.autograd.grad(o2, leaves, gy)
    a, b_ = run(True), run(False)
    for x, y in zip(a, b_):
        assert torch.equal(x, y)


def test_samediff_mixed_precision_step_links_match_plain_autograd():
    """Tiny BERT (bf16 compute, fp32 masters) through the optimizer's LinearFusion: one training step with the Links / transposed
    replicas / direct gradient accumulation must produce the same parameter update as the plain per-variable autograd path."""
    from deeplearning4j_amd.autodiff.optimize import GraphOptimizer
    from deeplearning4j_amd.autodiff.samediff import SameDiff, TrainingConfig
    from deeplearning4j_amd.models.bert import BertConfig, build_bert_classifier
    from deeplearning4j_amd.nd4j.learning import Sgd
    import deeplearning4j_amd.ops.linear as LIN
    cfg = BertConfig.tiny(vocab=96, hidden=128, layers=2, heads=2, intermediate=256, maxPositions=32)
    g = torch.Generator().manual_seed(0)
    ph = {"tokenIdx": torch.randint(0, cfg.vocabSize, (8, 32), generator=g).cuda(), "mask": torch.ones(8, 32).cuda(),
          "label": torch.eye(2)[torch.randint(0, 2, (8,), generator=g)].cuda()}

    def build(fuse):
        sd = SameDiff.create("cuda")
        build_bert_classifier(sd, cfg, 32, 2, computeDataType="BFLOAT16")
        if fuse:
            GraphOptimizer.optimize(sd)
            assert sum(o.opName == "fused_dense" for o in sd.ops.values()) == 7
        sd.setTrainingConfig(TrainingConfig.builder().updater(Sgd(0.5)).build())
        return sd

    a, b = build(False), build(True)
    names = a.trainableVariables()
    p0 = {n: a.getArrForVarName(n).tensor().clone() for n in names}
    real = LIN.set_links
    a.fit(ph); a.fit(ph)
    seen = []
    LIN.set_links = lambda l: (seen.append(l is not None), real(l))[1]
    try:
        b.fit(ph); b.fit(ph)                     # the second step runs on the flat vector + links
    finally:
        LIN.set_links = real
    assert any(seen)
    worst = 0.0
    for n in names:
        da = a.getArrForVarName(n).tensor() - p0[n]
        db = b.getArrForVarName(n).tensor() - p0[n]
        scale = max(float(da.abs().max()), 1e-6)
        worst = max(worst, float((da - db).abs().max()) / scale)
    assert worst < 8e-2, worst
