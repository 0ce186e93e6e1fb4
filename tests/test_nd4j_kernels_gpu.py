"""GPU numerics for the nd4j-level HIP kernels: legacy op families (legacy_ops.hip) vs the PyTorch fp32/fp64
formulation of each functor, wave64 threshold/bitmap codecs (codec.hip) vs the host codec, Philox fills
(random.hip) vs the host stream, multi-array average/accumulate, pullRows."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from deeplearning4j_amd.nd4j.ops import legacy as LG


def _ex():
    return LG.executioner()


def _close(a, b, rtol=2e-5, atol=2e-6):
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.dtype == torch.bool or not a.is_floating_point():
        return torch.equal(a.long(), b.long())
    return torch.allclose(a.double(), b.double(), rtol=rtol, atol=atol, equal_nan=True)


_POS_ONLY = {"Log", "Sqrt", "RSqrt", "ACosh", "ACoshDerivative", "Log1p", "LogX"}
_UNIT = {"ACos", "ASin", "ATanh"}


@pytest.mark.parametrize("family,table", [("transform_same", LG.TRANSFORM_SAME), ("transform_strict", LG.TRANSFORM_STRICT),
                                          ("transform_float", LG.TRANSFORM_FLOAT), ("transform_bool", LG.TRANSFORM_BOOL)])
def test_transforms_all_opnums(family, table):
    g = torch.Generator().manual_seed(1)
    base = torch.randn(33, 77, generator=g)
    ex = _ex()
    for opnum, name in table.items():
        x = base.clone()
        if name in _POS_ONLY:
            x = x.abs() + 1.1
        if name in _UNIT:
            x = x.clamp(-0.9, 0.9)
        extra = {"ClipByValue": [-0.5, 0.5], "ScaledTanh": [1.7, 0.6], "Affine": [2.0, -1.0], "Stabilize": [3.0], "StabilizeFP16": [3.0],
                 "SetRange": [0.0, 1.0], "CompareAndSetTransform": [0.0, 0.0, 2, 9.0], "MatchConditionBool": [0.5, 0.0, 3]}.get(name)
        fn = getattr(ex, "exec" + "".join(p.capitalize() for p in family.split("_")))
        ref = fn(opnum, x, None, extra)
        xg = x.cuda()
        out = fn(opnum, xg, None, extra)
        assert out.is_cuda and _close(out, ref, 1e-4, 1e-5), (family, name)
        # strided (transposed) input, fp64 and bf16 instantiations
        outT = fn(opnum, xg.t(), None, extra)
        assert _close(outT, ref.t(), 1e-4, 1e-5), (family, name, "strided")
        if family != "transform_bool":
            o64 = fn(opnum, x.double().cuda(), None, extra)
            r64 = fn(opnum, x.double(), None, extra)
            assert _close(o64, r64, 1e-9, 1e-10), (family, name, "fp64")
    xb = base.bfloat16().cuda()
    assert _close(ex.execTransformStrict(29, xb).float(), torch.tanh(base.bfloat16().float()).bfloat16().float(), 1e-2, 1e-2)


def test_pairwise_scalar_broadcast_all_opnums():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(19, 45, generator=g)
    y = torch.randn(19, 45, generator=g)
    y = torch.where(y.abs() < 0.2, torch.full_like(y, 0.7), y)
    ex = _ex()
    pos = {"Pow", "PowDerivative", "LogPoissonLossFull", "IGamma", "IGammac", "LogX"}
    extras = {"Axpy": [0.5], "CompareAndSet": [0.0, 0.0, 2], "CompareAndReplace": [0.0, 0.0, 3], "BinaryRelativeError": [0.3],
              "BinaryMinimumAbsoluteRelativeError": [0.3, 0.05]}
    for opnum, name in LG.PAIRWISE.items():
        xx, yy = (x.abs() + 0.5, y.abs() + 0.5) if name in pos else (x, y)
        ref = ex.execPairwiseTransform(opnum, xx, yy, None, extras.get(name))
        out = ex.execPairwiseTransform(opnum, xx.cuda(), yy.cuda(), None, extras.get(name))
        assert _close(out, ref, 2e-4, 1e-5), ("pairwise", name)
    for opnum, name in LG.PAIRWISE_BOOL.items():
        ext = [0.5, 0.0, 3] if name == "MatchCondition" else ([1e-3] if name == "Epsilon" else None)
        assert _close(ex.execPairwiseTransformBool(opnum, x.cuda(), y.cuda(), None, ext), ex.execPairwiseTransformBool(opnum, x, y, None, ext)), name
    xi, yi = torch.randint(-1000, 1000, (300,), generator=g, dtype=torch.int32), torch.randint(0, 8, (300,), generator=g, dtype=torch.int32)
    for opnum, name in LG.PAIRWISE_INT.items():
        assert _close(ex.execPairwiseTransformInt(opnum, xi.cuda(), yi.cuda()), ex.execPairwiseTransformInt(opnum, xi, yi)), name
    for opnum, name in LG.SCALAR.items():
        xx = x.abs() + 0.5 if name in pos | {"ReversePow"} else x
        sc = 1.5 if name not in ("CompareAndSet",) else 7.0
        ext = [0.0, 0.0, 2] if name == "CompareAndSet" else None
        assert _close(ex.execScalar(opnum, xx.cuda(), sc, None, ext), ex.execScalar(opnum, xx, sc, None, ext), 2e-4, 1e-5), ("scalar", name)
    row, col = torch.randn(45, generator=g).abs() + 0.5, torch.randn(19, generator=g).abs() + 0.5
    for opnum, name in LG.BROADCAST.items():
        xx = x.abs() + 0.5 if name in pos else x
        for vec, dim in ((row, [1]), (col, [0])):
            assert _close(ex.execBroadcast(opnum, xx.cuda(), vec.cuda(), None, dim), ex.execBroadcast(opnum, xx, vec, None, dim), 2e-4, 1e-5), (name, dim)
    # in-place + rank-4 strided operands (NHWC view of NCHW storage)
    a = torch.randn(4, 6, 5, 7, generator=g)
    ag = a.cuda().permute(0, 2, 3, 1)
    out = ex.execPairwiseTransform(0, ag, ag.contiguous(), ag)
    assert out.data_ptr() == ag.data_ptr() and _close(ag, 2 * a.permute(0, 2, 3, 1))


def test_reductions_all_opnums_and_two_stage():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(7, 130, 9, generator=g)
    x[0, 0, 0] = 0.0
    y = torch.randn(7, 130, 9, generator=g)
    ex = _ex()
    xg, yg = x.cuda(), y.cuda()
    for dims in (None, [1], [0, 2], [2]):
        for opnum, name in LG.REDUCE_FLOAT.items():
            xx, xxg = (x.abs() + 0.1, xg.abs() + 0.1) if "Entropy" in name else (x, xg)
            ext = [3.0] if name == "NormP" else None
            assert _close(ex.execReduceFloat(opnum, xxg, dims, False, ext), ex.execReduceFloat(opnum, xx, dims, False, ext), 1e-4, 1e-4), (name, dims)
        for opnum, name in LG.REDUCE_SAME.items():
            xs, xsg = (x[:, :12], xg[:, :12]) if name == "Prod" else (x, xg)
            assert _close(ex.execReduceSame(opnum, xsg, dims, True), ex.execReduceSame(opnum, xs, dims, True), 1e-4, 1e-4), (name, dims)
        for opnum, name in LG.REDUCE_BOOL.items():
            assert _close(ex.execReduceBool(opnum, xg, dims), ex.execReduceBool(opnum, x, dims)), (name, dims)
        for opnum, name in LG.REDUCE_LONG.items():
            ext = [0.5, 0.0, 3] if name == "MatchCondition" else None
            assert _close(ex.execReduceLong(opnum, xg, dims, False, ext), ex.execReduceLong(opnum, x, dims, False, ext)), (name, dims)
        for opnum, name in LG.REDUCE3.items():
            xa, ya = (x.abs(), y.abs()) if name == "JaccardDistance" else (x, y)
            assert _close(ex.execReduce3(opnum, xa.cuda(), ya.cuda(), dims), ex.execReduce3(opnum, xa, ya, dims), 1e-4, 1e-4), (name, dims)
        for opnum, name in LG.INDEX_REDUCE.items():
            ext = [1.0, 0.0, 3] if "Index" in name and name.startswith(("First", "Last")) else None
            assert _close(ex.execIndexReduce(opnum, xg, dims, False, ext), ex.execIndexReduce(opnum, x, dims, False, ext)), (name, dims)
        for opnum in LG.SUMMARY_STATS:
            for bc in (True, False):
                assert _close(ex.execSummaryStats(opnum, xg, dims, bc), ex.execSummaryStats(opnum, x, dims, bc), 1e-4, 1e-5), (opnum, dims, bc)
    # long full reduction -> segmented two-stage path; strided view; fp64; int64
    big = torch.randn(3_000_001, generator=g)
    assert abs(float(ex.execReduceSame(0, big.cuda())) - float(big.double().sum())) < 1e-2
    assert float(ex.execReduceSame(1, big.cuda())) == float(big.max())
    assert _close(ex.execReduceFloat(0, big.cuda().double()), big.double().mean(), 1e-12, 1e-12)
    m = torch.randn(2, 300_000, generator=g)
    assert _close(ex.execReduceFloat(3, m.cuda(), [1]), m.double().norm(dim=1).float(), 1e-5, 1e-4)
    assert _close(ex.execReduceSame(0, xg.permute(2, 0, 1), [1]), x.permute(2, 0, 1).sum(1), 1e-4, 1e-4)
    xi = torch.randint(-50, 50, (40, 33), generator=g)
    assert _close(ex.execReduceSame(0, xi.cuda(), [1]), xi.sum(1)) and _close(ex.execIndexReduce(0, xi.cuda(), [0]), xi.argmax(0))


def test_indarray_ops_run_on_gpu():
    from deeplearning4j_amd.nd4j.factory import Nd4j
    a = Nd4j.create([[1.0, 2, 3], [4, 5, 6]], device="cuda")
    assert a.tensor().is_cuda and a.add(1).mul(2).sumNumber() == 54 and a.mean(0).toDoubleVector() == [2.5, 3.5, 4.5]
    assert a.addRowVector(Nd4j.create([1.0, 1, 1], device="cuda")).maxNumber() == 7 and a.gt(3).castTo("INT").sumNumber() == 3
    assert abs(a.std(1).getDouble(1) - 1.0) < 1e-6 and a.argMax(1).toIntVector() == [2, 2]
    rows = Nd4j.pullRows(Nd4j.arange(12, device="cuda").reshape(4, 3), 1, [3, 0])
    assert rows.toDoubleMatrix() == [[9, 10, 11], [0, 1, 2]]
    with pytest.raises(IndexError):
        Nd4j.pullRows(Nd4j.arange(12, device="cuda").reshape(4, 3), 1, [4])
    xs = [Nd4j.rand(1000, 37, seed=s, device="cuda") for s in range(3)]
    ref = sum(x.tensor().clone() for x in xs) / 3
    tgt = Nd4j.zeros(1000, 37, device="cuda")
    Nd4j.averageAndPropagate(xs, tgt)
    assert all(torch.allclose(x.tensor(), ref, atol=1e-6) for x in xs) and torch.allclose(tgt.tensor(), ref, atol=1e-6)
    acc = Nd4j.zeros(1000, 37, device="cuda")
    Nd4j.accumulate(acc, xs)
    assert torch.allclose(acc.tensor(), 3 * ref, atol=1e-5)


def test_codecs_device_matches_host():
    from deeplearning4j_amd.nd4j.compression import ThresholdCompression as TC
    g = torch.Generator().manual_seed(5)
    for n in (1, 63, 64, 4096, 4097, 1_000_003):
        x = torch.randn(n, generator=g) * 0.01
        tau = 0.015
        xh, xd = x.clone(), x.clone().cuda()
        eh, ed = TC.thresholdEncode(xh, tau), TC.thresholdEncode(xd, tau)
        if eh is None:
            assert ed is None
            continue
        assert torch.equal(eh, ed.cpu()), n                    # identical message incl. index order
        assert torch.equal(xh, xd.cpu())                       # identical residual
        z = torch.zeros(n, device="cuda")
        TC.thresholdDecode(ed, z)
        assert torch.allclose(z.cpu() + xh, x, atol=1e-7)
        bh, bd = x.clone(), x.clone().cuda()
        mh, md = TC.bitmapEncode(bh, tau), TC.bitmapEncode(bd, tau)
        assert torch.equal(mh, md.cpu()) and torch.equal(bh, bd.cpu()), n
        zz = torch.zeros(n, device="cuda")
        TC.bitmapDecode(md, zz)
        assert torch.allclose(zz.cpu() + bh, x, atol=1e-7)
    big = (torch.randn(2_000_000, generator=g) * 0.01).cuda()
    ref = big.clone()
    capped = TC.thresholdEncode(big, 0.01, boundary=1000)
    assert TC.header(capped)["count"] == 1000
    first = torch.nonzero(ref.abs() >= 0.01).reshape(-1)[:1000]
    assert torch.equal(capped[4:].long().abs() - 1, first)
    untouched = torch.ones_like(ref, dtype=torch.bool)
    untouched[first] = False
    assert torch.equal(big[untouched], ref[untouched])


def test_random_device_matches_host_stream():
    from deeplearning4j_amd.nd4j import rng as R
    for op, args in (("UniformDistribution", [-2.0, 3.0]), ("GaussianDistribution", [1.0, 0.5]), ("BernoulliDistribution", [0.3])):
        zc, zg = torch.empty(100_001), torch.empty(100_001, device="cuda")
        R.exec_random(op, R.Random(42), zc, None, None, args)
        R.exec_random(op, R.Random(42), zg, None, None, args)
        assert torch.allclose(zc, zg.cpu(), rtol=1e-5, atol=1e-6), op
    z = torch.empty(400_000, device="cuda")
    r = R.Random(9)
    R.exec_random("TruncatedNormalDistribution", r, z, None, None, [0.0, 0.5])
    assert float(z.abs().max()) <= 1.0 + 1e-6 and abs(float(z.mean())) < 5e-3 and 0.4 < float(z.std()) < 0.46
    R.exec_random("BinomialDistribution", r, z, None, None, [10, 0.25])
    assert abs(float(z.mean()) - 2.5) < 2e-2 and abs(float(z.var()) - 1.875) < 5e-2
    R.exec_random("ExponentialDistribution", r, z, None, None, [4.0])
    assert abs(float(z.mean()) - 0.25) < 3e-3
    R.exec_random("PoissonDistribution", r, z, None, None, [3.0])
    assert abs(float(z.mean()) - 3.0) < 2e-2 and abs(float(z.var()) - 3.0) < 6e-2
    R.exec_random("PoissonDistribution", r, z, None, None, [100.0])
    assert abs(float(z.mean()) - 100.0) < 0.2
    R.exec_random("GammaDistribution", r, z, None, None, [2.5, 2.0])
    assert abs(float(z.mean()) - 1.25) < 1e-2 and abs(float(z.var()) - 0.625) < 2e-2
    R.exec_random("GammaDistribution", r, z, None, None, [0.5, 1.0])
    assert abs(float(z.mean()) - 0.5) < 1e-2
    R.exec_random("LogNormalDistribution", r, z, None, None, [0.0, 0.25])
    assert abs(float(z.log().mean())) < 3e-3
    x = torch.ones(400_000, device="cuda")
    R.exec_random("DropOutInverted", r, z, x, None, [0.8])
    assert abs(float(z.mean()) - 1.0) < 5e-3 and abs(float((z == 0).float().mean()) - 0.2) < 5e-3
    src, probs = torch.tensor([10.0, 20.0, 30.0], device="cuda"), torch.tensor([0.2, 0.5, 0.3], device="cuda")
    R.exec_random("Choice", r, z, src, probs, None)
    assert abs(float((z == 20.0).float().mean()) - 0.5) < 5e-3
    zb = torch.empty(10_000, device="cuda", dtype=torch.bfloat16)
    R.exec_random("UniformDistribution", r, zb, None, None, [0.0, 1.0])
    assert 0.45 < float(zb.float().mean()) < 0.55
    zd = torch.empty(10_000, device="cuda", dtype=torch.float64)
    R.exec_random("GaussianDistribution", r, zd, None, None, [0.0, 1.0])
    assert abs(float(zd.mean())) < 5e-2


def test_custom_op_platform_helpers_run_hip_paths():
    from deeplearning4j_amd.nd4j.ops.custom import exec_op
    g = torch.Generator().manual_seed(7)
    x = torch.randn(64, 100, generator=g).cuda()
    assert torch.allclose(exec_op("softmax", [x])[0], torch.softmax(x, 1), atol=1e-6)
    img = torch.randn(4, 64, 14, 14, generator=g).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).cuda().bfloat16()
    b = torch.randn(64, generator=g).cuda()
    out = exec_op("conv2d", [img, w, b], iArgs=[3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 1])[0]
    ref = torch.nn.functional.conv2d(img.float(), w.float(), b, 1, 1)
    assert (out.float() - ref).abs().max() < 3e-2 * max(1.0, float(ref.abs().max()))
    xf = torch.randn(4, 32, 9, 9, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    mp = exec_op("maxpool2d", [xf], iArgs=[3, 3, 2, 2, 1, 1, 1, 1, 0, 0, 0])[0]
    assert torch.allclose(mp, torch.nn.functional.max_pool2d(xf, 3, 2, 1))
    gamma, beta, mean, var = [torch.rand(32, generator=g).cuda() + 0.5 for _ in range(4)]
    bn = exec_op("batchnorm", [xf, mean, var, gamma, beta], iArgs=[1, 1, 1], tArgs=[1e-5])[0]
    ref = (xf - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    assert torch.allclose(bn, ref, atol=1e-4)


def test_launch_dims_overrides_are_honoured_by_the_kernels():
    """Environment.setLaunchDims reaches the kernels (LaunchDims.cu analogue): the legacy transform family and BatchNorm launch with
    the capped grid and still produce the right values (grid-stride kernels)."""
    from deeplearning4j_amd.nd4j.environment import Environment
    from deeplearning4j_amd.nd4j.ops import legacy as LG
    from deeplearning4j_amd.ops import batchnorm as BN
    env = Environment.getInstance()
    x = torch.randn(1 << 20, device="cuda")
    lg = LG.executioner()
    opn = next(k for k, v in LG.TRANSFORM_STRICT.items() if v == "Tanh")
    ref = torch.tanh(x)
    try:
        y0 = lg.execTransformStrict(opn, x, None, None)
        free = env.kernelLaunchDims("transformScan")
        assert free is not None and free[2] > 64
        env.setLaunchDims("transformScan", 7, 128)
        y1 = lg.execTransformStrict(opn, x, None, None)
        assert env.kernelLaunchDims("transformScan") == (7, 128, 7)
        assert torch.allclose(y0, ref, atol=1e-6) and torch.allclose(y1, ref, atol=1e-6)
        env.setLaunchDims("batch_norm", 3, 256)
        a = torch.randn(4096, 64, device="cuda")
        sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda")
        out = BN.apply(a, sc, sh, relu=True)
        assert env.kernelLaunchDims("batch_norm")[2] == 3
        assert torch.allclose(out, torch.relu(a * sc + sh), atol=1e-5)
    finally:
        env.setLaunchDims("transformScan", 0, 0)
        env.setLaunchDims("batch_norm", 0, 0)
