"""Aggregate batches on the GPU: one launch of native/csrc/aggregates.hip per batch (axpy / dot / small column-major GEMMs) and the
batched word2vec kernel for skip-gram batches, against per-item torch math. Reference: loops/aggregates.h, NativeOps.h:931-942."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def test_blas_aggregate_batches_are_single_launches_and_match_torch():
    from deeplearning4j_amd.nd4j.ops import aggregates as A
    from deeplearning4j_amd.nd4j.ops.custom import OpExecutioner
    ex = OpExecutioner()
    g = torch.Generator().manual_seed(0)
    sizes = [1, 63, 64, 65, 300, 1000, 4097] * 20
    xs = [torch.randn(n, generator=g).to(dev()) for n in sizes]
    ys = [torch.randn(n, generator=g).to(dev()) for n in sizes]
    want = [y + (0.1 * (k + 1)) * x for k, (x, y) in enumerate(zip(xs, ys))]
    A.STATS.update(launches=0, items_native=0, items_generic=0)
    ex.exec(A.Batch([A.AggregateAxpy(x, y, 0.1 * (k + 1)) for k, (x, y) in enumerate(zip(xs, ys))]))
    assert A.STATS["launches"] == 1 and A.STATS["items_native"] == len(sizes) and A.STATS["items_generic"] == 0
    for y, w in zip(ys, want):
        assert torch.allclose(y, w, atol=1e-5, rtol=1e-5)
    dots = [A.AggregateDot(x, y) for x, y in zip(xs, ys)]
    ex.exec(A.Batch(dots))
    assert A.STATS["launches"] == 2
    for d, x, y in zip(dots, xs, ys):
        ref = float(x.double() @ y.double())
        assert abs(d.getFinalResult() - ref) <= 1e-3 * max(1.0, abs(ref)), (d.getFinalResult(), ref)
    items, checks = [], []
    for k, (M, N, K) in enumerate([(5, 7, 4), (16, 16, 16), (33, 17, 50), (1, 9, 3), (64, 48, 100), (20, 1, 31)] * 4):
        ta, tb = (111, 112)[k % 2], (111, 112)[(k // 2) % 2]
        ar, ac = (K, M) if ta == 112 else (M, K)
        br, bc = (N, K) if tb == 112 else (K, N)
        lda, ldb, ldc = ar + (k % 3), br + 1, M + 2
        Am, Bm, Cm = (torch.randn(n, generator=g).to(dev()) for n in (lda * ac, ldb * bc, ldc * N))
        a = torch.as_strided(Am, (ar, ac), (1, lda))
        b = torch.as_strided(Bm, (br, bc), (1, ldb))
        beta = 0.0 if k % 5 == 0 else -0.5
        ref = 1.25 * ((a.t() if ta == 112 else a) @ (b.t() if tb == 112 else b)) + beta * torch.as_strided(Cm, (M, N), (1, ldc))
        items.append(A.AggregateGEMM(102, ta, tb, M, N, K, 1.25, Am, lda, Bm, ldb, beta, Cm, ldc))
        checks.append((Cm, M, N, ldc, ref.clone(), Cm.clone()))
    ex.exec(A.Batch(items))
    assert A.STATS["launches"] == 3 and A.STATS["items_generic"] == 0
    for Cm, M, N, ldc, ref, before in checks:
        assert torch.allclose(torch.as_strided(Cm, (M, N), (1, ldc)), ref, atol=2e-4, rtol=1e-4)
        pad = torch.ones(Cm.numel(), dtype=torch.bool, device=dev())
        torch.as_strided(pad, (M, N), (1, ldc)).fill_(False)
        assert torch.equal(Cm[pad], before[pad])
    # an operand the kernel cannot take (fp64) runs the generic path and says so
    x64, y64 = torch.ones(8, device=dev(), dtype=torch.float64), torch.zeros(8, device=dev(), dtype=torch.float64)
    ex.exec(A.AggregateAxpy(x64, y64, 2.0))
    assert A.STATS["items_generic"] == 1 and torch.equal(y64, 2 * x64)


def test_skipgram_aggregate_batch_runs_the_batched_word2vec_kernel():
    from deeplearning4j_amd.nd4j.ops import aggregates as A
    from deeplearning4j_amd.nd4j.ops.custom import OpExecutioner
    ex = OpExecutioner()
    g = torch.Generator().manual_seed(1)
    V, D = 50, 64
    syn0 = (torch.randn(V, D, generator=g) * 0.1)
    syn1 = (torch.randn(V, D, generator=g) * 0.1)
    syn1n = (torch.randn(V, D, generator=g) * 0.1)
    neg = torch.randint(0, V, (1000,), generator=g, dtype=torch.int32)

    def items(t0, t1, t1n, ng):
        # distinct target rows and code points per item: the batch result does not depend on the order items are applied in
        return [A.AggregateSkipGram(t0, t1, t1n, None, ng, w, [10 + 2 * w, 11 + 2 * w], [0, 1], 0, 0, D, 0.025, 7 + w, V) for w in range(5)]

    c0, c1, c1n = syn0.clone(), syn1.clone(), syn1n.clone()
    ex.exec(A.Batch(items(c0, c1, c1n, neg)))
    g0, g1, g1n = syn0.to(dev()), syn1.to(dev()), syn1n.to(dev())
    OpExecutioner.resetHelperStatistics()
    ex.exec(A.Batch(items(g0, g1, g1n, neg.to(dev()))))
    assert OpExecutioner.helperStatistics()["hit"].get("skipgram", 0) == 1          # one batched kernel call for the five aggregates
    assert torch.allclose(g0.cpu(), c0, atol=1e-5) and torch.allclose(g1.cpu(), c1, atol=1e-5)
    assert not torch.allclose(g0.cpu(), syn0)
