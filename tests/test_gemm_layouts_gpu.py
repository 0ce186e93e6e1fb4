"""Every operand layout of the in-tree MFMA GEMM (native/csrc/gemm_trans.h: NN / TN / TT next to the native NT form, strided batches,
split-K) against fp32 torch matmuls. Reference coverage: libnd4j/include/helpers/cuda_off/MmulHelper.cu:255-341 (all transposes to
cuBLAS), helpers/cuda/batched_gemm.cu:145-193."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _operands(M, N, K, ta, tb, dt, batch=None, pad=0):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    lead = () if batch is None else (batch,)

    def make(rows, cols, transposed):
        if transposed:                      # stored [cols, rows + pad], viewed as [rows, cols]
            t = torch.randn(*lead, cols, rows + pad, generator=g).to(dev()).to(dt)
            return t[..., :rows].transpose(-1, -2)
        t = torch.randn(*lead, rows, cols + pad, generator=g).to(dev()).to(dt)
        return t[..., :cols]
    return make(M, K, ta), make(N, K, tb)


@pytest.mark.parametrize("dt,tol", [(torch.bfloat16, 2e-2), (torch.float16, 4e-3), (torch.float32, 2e-4)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_all_four_layouts_match_torch(dt, tol, ta, tb):
    from deeplearning4j_amd.ops import gemm as G
    ch = 4 if dt == torch.float32 else 8
    for (M, N, K), tile in (((256, 256, 256), 0), ((136, 72, 200), 1), ((64, 64, 64), 2), ((40, 264, 1024), 3), ((1000, 520, 96), None),
                            ((8, 8, 32), 2)):
        for pad in (0, ch):
            a, b = _operands(M, N, K, ta, tb, dt, pad=pad)
            assert G.supported_ex(a, b), (M, N, K, ta, tb, pad)
            got = G.gemm(a, b, tile=tile, splits=1)
            ref = a.float() @ b.float().transpose(-1, -2)
            scale = float(ref.abs().max())
            assert got.dtype == dt and got.shape == ref.shape
            err = float((got.float() - ref).abs().max())
            assert err <= tol * scale + 1e-5, (M, N, K, ta, tb, pad, tile, err, scale)


def test_epilogue_batch_and_split_k_on_transposed_operands():
    from deeplearning4j_amd.ops import gemm as G
    # bias + activation + residual on the NN form (row-major x @ w)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 192, generator=g).to(dev()).bfloat16()
    w = torch.randn(192, 136, generator=g).to(dev()).bfloat16()
    bias = torch.randn(136, generator=g).to(dev())
    res = torch.randn(300, 136, generator=g).to(dev()).bfloat16()
    got = G.matmul(x, w, bias=bias, act="relu", residual=res)
    ref = torch.relu(x.float() @ w.float() + bias) + res.float()
    assert float((got.float() - ref).abs().max()) <= 3e-2 * float(ref.abs().max())
    # strided batch, TN per batch item: scores^T-like products
    a, b = _operands(72, 88, 160, True, True, torch.bfloat16, batch=5)
    got = G.gemm(a, b, out_dtype=torch.float32)
    ref = a.float() @ b.float().transpose(-1, -2)
    assert got.shape == (5, 72, 88) and float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    # dense weight gradient shape: dW[nOut, nIn] = dz^T . x, K = tokens, both operands K-major, split over K with fp32 atomics
    for dt, tol in ((torch.bfloat16, 2e-2), (torch.float32, 3e-4)):
        dz = torch.randn(4096, 264, generator=g).to(dev()).to(dt)
        xx = torch.randn(4096, 520, generator=g).to(dev()).to(dt)
        ref = dz.float().t() @ xx.float()
        for splits in (None, 1, 7):
            got = G.gemm(dz.t(), xx.t(), out_dtype=torch.float32, splits=splits)
            assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()), (dt, splits)
        acc = torch.ones(264, 520, device=dev())
        G.gemm(dz.t(), xx.t(), out=acc, splits=4, accum=True)                    # += into the gradient buffer
        assert float((acc - 1 - ref).abs().max()) <= tol * float(ref.abs().max())
    # host-side refusals
    assert not G.supported_ex(x[:, ::2], w.t())                                   # neither dim dense
    assert not G.supported_ex(x, torch.randn(136, 100, device=dev()).bfloat16())  # K mismatch
    assert not G.supported_ex(torch.randn(60, 33, device=dev()).bfloat16().t(), torch.randn(64, 60, device=dev()).bfloat16())  # M % 8 with K-major A


def test_matmul_and_batched_gemm_ops_take_every_layout_without_copies(monkeypatch):
    """The custom ops reach the layout-aware kernel: no transpose / contiguous copy and no library GEMM inside the route."""
    from deeplearning4j_amd.nd4j.ops import cost_model
    from deeplearning4j_amd.nd4j.ops.custom import exec_op, OpExecutioner
    monkeypatch.setattr(cost_model, "ENABLED", False)
    g = torch.Generator().manual_seed(9)
    A = torch.randn(264, 520, generator=g).to(dev()).bfloat16()          # used as [M, K] or as [K, M]
    for ta in (0, 1):
        for tb in (0, 1):
            M, K = (520, 264) if ta else (264, 520)
            B = torch.randn(*((136, K) if tb else (K, 136)), generator=g).to(dev()).bfloat16()
            ref = (A.float().t() if ta else A.float()) @ (B.float().t() if tb else B.float())
            OpExecutioner.resetHelperStatistics()
            real = torch.matmul
            monkeypatch.setattr(torch, "matmul", lambda *a, **k: (_ for _ in ()).throw(AssertionError("library GEMM")))
            monkeypatch.setattr(torch.Tensor, "contiguous", lambda self, *a, **k: (_ for _ in ()).throw(AssertionError("copy")))
            try:
                got = exec_op("matmul", [A, B], iArgs=[ta, tb, 0], tArgs=[0.5, 0.0])[0]
            finally:
                monkeypatch.undo()
                monkeypatch.setattr(cost_model, "ENABLED", False)
            assert real is torch.matmul
            assert OpExecutioner.helperStatistics()["hit"].get("matmul") == 1
            assert got.shape == (M, 136) and float((got.float() - 0.5 * ref).abs().max()) <= 2e-2 * float(ref.abs().max()), (ta, tb)
    # batched_gemm: equally spaced operands -> one strided-batch launch; separate arrays -> one launch per pair
    bufA = torch.randn(5, 72, 160, generator=g).to(dev()).bfloat16()
    bufB = torch.randn(5, 160, 88, generator=g).to(dev()).bfloat16()
    alpha, beta = torch.tensor(2.0, device=dev()), torch.tensor(0.0, device=dev())
    OpExecutioner.resetHelperStatistics()
    outs = exec_op("batched_gemm", [alpha, beta] + [bufA[k] for k in range(5)] + [bufB[k] for k in range(5)], iArgs=[0, 0, 72, 88, 160, 72, 160, 72, 5])
    assert OpExecutioner.helperStatistics()["hit"].get("batched_gemm") == 1 and len(outs) == 5
    for k in range(5):
        ref = 2.0 * bufA[k].float() @ bufB[k].float()
        assert float((outs[k].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    loose = [torch.randn(160, 72, generator=g).to(dev()).bfloat16() for _ in range(3)]          # transA form, separately allocated
    looseB = [torch.randn(88, 160, generator=g).to(dev()).bfloat16() for _ in range(3)]         # transB form
    outs = exec_op("batched_gemm", [alpha, beta] + loose + looseB, iArgs=[1, 1, 72, 88, 160, 160, 88, 72, 3])
    for k in range(3):
        ref = 2.0 * loose[k].float().t() @ looseB[k].float().t()
        assert float((outs[k].float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 200), (1000, 1300, 72), (2048, 2048, 1024), (300, 260, 8)])
def test_pingpong_deep_loop_matches_fp32_and_the_drain_form_bitwise(M, N, K):
    """256 x 256 ping-pong GEMM with half-tile staging and counted vmcnt waits (tile 10, gemm_core.h mainloop_pingpong_deep) against fp32
    PyTorch; the MFMA order is the one of the drain-per-tile loop (tile 4), so repeated runs must agree with it BITWISE - a staged
    half-tile read before it landed (or re-staged too early) shows up as a differing run. Tails in M, N and K, K < one tile."""
    from deeplearning4j_amd.ops import gemm as G
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = torch.randn(N, K, device="cuda", generator=g).bfloat16()
    ref = x.float() @ w.float().t()
    y4 = G.gemm_nt(x, w, tile=4)
    for _ in range(8):
        y10 = G.gemm_nt(x, w, tile=10)
        assert torch.equal(y10, y4)
    assert float((y10.float() - ref).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("act", [None, "relu", "gelu", "gelu_erf", "tanh", "sigmoid"])
@pytest.mark.parametrize("tile", [2, 0, 10])
def test_every_epilogue_activation_body_on_small_and_pingpong_tiles(act, tile):
    """gemm_epilogue.h has one straight-line body per activation kind (linear / ReLU share the branch-free one): each of them, with bias
    and residual, on a 64 x 64 tile, the 128 x 128 tile and the 256 x 256 ping-pong tile."""
    from deeplearning4j_amd.ops import gemm as G
    g = torch.Generator(device="cuda").manual_seed(7)
    M, N, K = 520, 384, 192
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    z = x.float() @ w.float().t() + b
    f = {None: lambda t: t, "relu": torch.relu, "gelu": lambda t: torch.nn.functional.gelu(t, approximate="tanh"),
         "gelu_erf": torch.nn.functional.gelu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act]
    ref = f(z).bfloat16().float() + r.float()
    y = G.gemm_nt(x, w, bias=b, act=act, residual=r, tile=tile)
    assert float((y.float() - ref).abs().max()) <= 3e-2 * max(1.0, float(ref.abs().max()))
