"""Op-definition DSL + generators (deeplearning4j_amd/codegen). Reference: codegen/op-codegen tests (OpBuilderTest.kt, DocsGeneratorTest) and
the generated namespaces' own tests (platform-tests/.../SDLinalgTest, TestNamespaces)."""
import os

import pytest
import torch

from deeplearning4j_amd.autodiff.samediff import SameDiff
from deeplearning4j_amd.codegen import Count, DataType, Namespace, all_namespaces, generate_docs, generate_eager, generate_samediff
from deeplearning4j_amd.codegen.generator import check_registry, generate_all
from deeplearning4j_amd.nd4j.factory import Nd4j

def _tensor(x):
    f = getattr(x, "tensor", None)
    return x if f is None else (f() if callable(f) else f)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_checked_in_generated_files_are_current_and_ops_exist():
    nss = all_namespaces()
    assert {"bitwise", "linalg", "random", "image"} <= set(nss)
    assert generate_all(ROOT, check=True) == []                 # regenerate == what is committed
    for ns in nss.values():
        if ns.name.startswith("_test"):
            continue
        assert check_registry(ns) == [], ns.name


def test_dsl_validation_and_generated_source_shape():
    ns = Namespace("_test_ns", doc="scratch", className="Scratch")
    (ns.op("clip", libnd4j="clipbyvalue", doc="Clip values.").input("x", DataType.NUMERIC, "input").arg("lo", DataType.FLOATING_POINT, "lower bound")
       .arg("hi", DataType.FLOATING_POINT, "upper bound", default=1.0).output("out", DataType.NUMERIC, "clipped"))
    (ns.op("sumAlong", libnd4j="reduce_sum", doc="Sum over dimensions.").input("x", DataType.NUMERIC, "input")
       .arg("keepDims", DataType.BOOL, "keep reduced dimensions", default=False).arg("dims", DataType.INT, "dimensions", count=Count.AtLeast(0))
       .output("out", DataType.NUMERIC, "sums").alias("total"))
    (ns.op("pick", libnd4j="Where", doc="Select.").input("cond", DataType.BOOL, "condition").input("a", DataType.NUMERIC, "if true", optional=True)
       .input("b", DataType.NUMERIC, "if false", optional=True).output("out", DataType.NUMERIC, "selection"))
    src = generate_samediff(ns)
    assert "class GenSDScratch:" in src and "def clip(self, x, lo, hi=1.0, name=None):" in src and "tArgs=[lo, hi]" in src
    assert "def sumAlong(self, x, keepDims=False, *dims, name=None):" in src and "iArgs=[v for v in _dims(dims)]" in src and "bArgs=[keepDims]" in src
    assert "total = sumAlong" in src and "[cond] + ([a] if a is not None else []) + ([b] if b is not None else [])" in src
    compile(src, "gen_sd", "exec")
    compile(generate_eager(ns), "gen_nd", "exec")
    md = generate_docs(ns)
    assert "## clip" in md and "* **hi** (FLOATING_POINT) - upper bound - default = 1.0" in md and "aliases: `total`" in md
    bad = Namespace("_test_bad")
    bad.op("noOutput").input("x", DataType.NUMERIC, "x")
    with pytest.raises(ValueError, match="at least one output"):
        bad.validate()
    bad2 = Namespace("_test_bad2")
    bad2.op("order").arg("a", DataType.INT, "a", default=1).arg("b", DataType.INT, "b").output("o", DataType.INT, "o")
    with pytest.raises(ValueError, match="follows one with a default"):
        bad2.validate()
    with pytest.raises(ValueError):
        Namespace("_test_bad3").op("x").arg("mode", DataType.ENUM, "no values")
    with pytest.raises(ValueError):
        Namespace("_test_bad4").op("x").input("not an identifier", DataType.INT, "")
    for n in ("_test_ns", "_test_bad", "_test_bad2", "_test_bad3", "_test_bad4"):
        Namespace._ALL.pop(n, None)


def test_generated_samediff_namespaces_build_and_run_graphs():
    sd = SameDiff.create()
    a = sd.constant("a", torch.tensor([[4.0, 1.0], [1.0, 3.0]]))
    b = sd.constant("b", torch.tensor([[1.0], [2.0]]))
    la = sd.linalg
    x = la.solve(a, b)                                   # hand-written override
    band = la.matrixBandPart(a, 0, 0)                    # generated only in name order (minLower, maxUpper)
    tri = la.tri(3, 4, -1)
    s, u, v = la.svd(a, False, True)
    only_s = la.svd(a, computeUV=False)
    i1 = sd.constant("i1", torch.tensor([0b1100, 0b1010], dtype=torch.int32))
    i2 = sd.constant("i2", torch.tensor([0b1010, 0b0110], dtype=torch.int32))
    bw = sd.bitwise
    r_and, r_tog = bw.and_(i1, i2), bw.toggleBits(i1)
    assert bw.bitsAnd is not None and type(bw).and_ is not None
    img = sd.constant("img", torch.arange(1, 10, dtype=torch.float32).reshape(1, 3, 3, 1))
    area = sd.image.resizeArea(img, 6, 6)
    rnd = sd.random.uniform(2.0, 3.0, "FLOAT", 4, 5)
    out = sd.output({}, x.name, band.name, tri.name, s.name, only_s.name, r_and.name, r_tog.name, area.name, rnd.name)
    t = lambda v_: _tensor(out[v_.name])
    torch.testing.assert_close(t(x), torch.linalg.solve(torch.tensor([[4.0, 1.0], [1.0, 3.0]]), torch.tensor([[1.0], [2.0]])))
    torch.testing.assert_close(t(band), torch.tensor([[4.0, 0.0], [0.0, 3.0]]))
    torch.testing.assert_close(t(tri).float(), torch.tril(torch.ones(3, 4), -1))
    torch.testing.assert_close(t(s), t(only_s))
    assert t(r_and).tolist() == [0b1000, 0b0010] and t(r_tog).tolist() == [~0b1100, ~0b1010]
    assert t(area).shape == (1, 6, 6, 1) and float(t(area)[0, 0, 1, 0]) == 1.0 and float(t(area)[0, 5, 5, 0]) == 9.0
    r = t(rnd)
    assert r.shape == (4, 5) and float(r.min()) >= 2.0 and float(r.max()) <= 3.0


def test_generated_eager_namespaces():
    a = Nd4j.create([[4.0, 1.0], [1.0, 3.0]])
    inv = Nd4j.linalg().matrixInverse(a)
    torch.testing.assert_close(_tensor(inv), torch.linalg.inv(_tensor(a)))
    q, r = Nd4j.linalg().qr(a)
    torch.testing.assert_close(_tensor(q) @ _tensor(r), _tensor(a))
    mm = Nd4j.linalg().mmul(a, a, transposeB=True)
    torch.testing.assert_close(_tensor(mm), _tensor(a) @ _tensor(a).t())
    x = torch.tensor([1, 2, 4], dtype=torch.int32)
    assert _tensor(Nd4j.bitwise().leftShift(x, torch.tensor([1, 1, 2], dtype=torch.int32))).tolist() == [2, 4, 16]
    assert _tensor(Nd4j.bitwise().xor(x, x)).tolist() == [0, 0, 0]
    n = Nd4j.random().normal(0.0, 1.0, "DOUBLE", [1000])
    assert _tensor(n).dtype == torch.float64 and _tensor(n).shape == (1000,) and abs(float(_tensor(n).mean())) < 0.2
    hsv = Nd4j.image().rgbToHsv(torch.tensor([[1.0, 0.0, 0.0]]))
    torch.testing.assert_close(_tensor(hsv), torch.tensor([[0.0, 1.0, 1.0]]))
    crop = Nd4j.image().cropAndResize(torch.rand(1, 8, 8, 1), torch.tensor([[0.0, 0.0, 1.0, 1.0]]), torch.tensor([0], dtype=torch.int32),
                                      torch.tensor([4, 4], dtype=torch.int32), method="NEAREST")
    assert _tensor(crop).shape == (1, 4, 4, 1)
    # ops outside the generated surface are still reachable by registry name
    assert _tensor(Nd4j.math().reduce_sum(torch.tensor([1.0, 2.0]))).tolist() == 3.0


def test_generated_loss_namespace_matches_reference_math():
    """codegen/ops/loss.py: inputs reach the native op as (predictions, weights, labels), an absent weights input becomes the constant 1.0,
    the reduction enum maps to libnd4j's mode number; SameDiff and eager surfaces agree with plain torch."""
    from deeplearning4j_amd.autodiff.samediff import SameDiff
    from deeplearning4j_amd.nd4j.factory import Nd4j
    g = torch.Generator().manual_seed(0)
    lab, pred = torch.rand(6, 5, generator=g), torch.rand(6, 5, generator=g)
    w = torch.rand(6, 1, generator=g)
    sd = SameDiff.create("cpu")
    L, P, W = sd.constant("l", lab), sd.constant("p", pred), sd.constant("w", w)
    mse = sd.loss.meanSquaredError(L, P)
    mse_w = sd.loss.meanSquaredError(L, P, W, "SUM")
    hub = sd.loss.huberLoss(L, P, None, 3, 0.3)
    out = sd.output({}, [mse.name, mse_w.name, hub.name])
    t = lambda n: _tensor(out[n])
    assert abs(float(t(mse.name)) - float(((lab - pred) ** 2).mean())) < 1e-6
    assert abs(float(t(mse_w.name)) - float((((lab - pred) ** 2) * w).sum())) < 1e-5
    d = (lab - pred).abs()
    ref_h = torch.where(d <= 0.3, 0.5 * d * d, 0.3 * d - 0.5 * 0.09).mean()
    assert abs(float(t(hub.name)) - float(ref_h)) < 1e-6
    e = Nd4j.loss().absoluteDifference(lab, pred, None, "NONE")
    assert torch.allclose(_tensor(e), (lab - pred).abs(), atol=1e-6)
    logits, cls = torch.randn(4, 7, generator=g), torch.tensor([1, 0, 6, 3])
    sp = Nd4j.loss().sparseSoftmaxCrossEntropy(logits, cls)
    assert torch.allclose(_tensor(sp), torch.nn.functional.cross_entropy(logits, cls, reduction="none"), atol=1e-5)


# ------------------------------------------------------------------------------------------------ math / nn / cnn / rnn / base namespaces
def test_all_ten_namespaces_are_defined_generated_and_current():
    import os
    nss = all_namespaces()
    assert sorted(nss) == ["base", "bitwise", "cnn", "image", "linalg", "loss", "math", "nn", "random", "rnn"]
    for n, ns in nss.items():
        assert check_registry(ns) == [], n
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert generate_all(root, None, check=True) == []          # python -m deeplearning4j_amd.codegen --check


def _t(x):
    return x.tensor() if hasattr(x, "tensor") else x


def test_generated_eager_base_and_math_match_torch():
    import torch
    from deeplearning4j_amd.autodiff.generated.nd_base import NDBase
    from deeplearning4j_amd.autodiff.generated.nd_math import NDMath
    torch.manual_seed(0)
    b, m = NDBase(), NDMath()
    x, y = torch.randn(3, 4, 5), torch.randn(3, 4, 5)
    assert torch.equal(_t(b.concat([x, y], 1)), torch.cat([x, y], 1)) and torch.equal(_t(b.stack([x, y], 0)), torch.stack([x, y]))
    assert [tuple(t.shape) for t in b.unstack(x, 1, 4)] == [(3, 5)] * 4
    assert torch.equal(_t(b.reshape(x, 12, 5)), x.reshape(12, 5)) and torch.equal(_t(b.permute(x, 2, 0, 1)), x.permute(2, 0, 1))
    assert torch.equal(_t(b.expandDims(x, 1)), x.unsqueeze(1)) and torch.equal(_t(b.tile(x, 1, 2, 1)), x.repeat(1, 2, 1))
    assert torch.allclose(_t(b.sum(x, True, 1)), x.sum(1, keepdim=True)) and torch.allclose(_t(b.mean(x, False, 0, 2)), x.mean((0, 2)))
    assert torch.allclose(_t(b.norm2(x, False, 2)), x.norm(dim=2)) and torch.allclose(_t(b.variance(x, False, True, 1)), x.var(1, unbiased=True), atol=1e-6)
    assert torch.equal(_t(b.argmax(x, False, 2)), x.argmax(2)) and torch.allclose(_t(b.cumsum(x, False, False, 1)), x.cumsum(1), atol=1e-6)
    assert torch.equal(_t(b.gt(x, y)), x > y) and torch.equal(_t(b.lte(x, y)), x <= y)
    idx = torch.tensor([2, 0])
    assert torch.equal(_t(b.gather(x, idx, 1)), x[:, idx]) and torch.equal(_t(b.oneHot(idx, 3)), torch.nn.functional.one_hot(idx, 3).float())
    ids = torch.tensor([0, 0, 1])
    assert torch.allclose(_t(b.segmentSum(x, ids)), torch.stack([x[0] + x[1], x[2]])) and tuple(_t(b.unsortedSegmentMax(x, ids, 2)).shape) == (2, 4, 5)
    assert torch.equal(_t(b.shape(x)), torch.tensor([3, 4, 5])) and int(_t(b.size(x))) == 60
    assert torch.allclose(_t(b.mmul(x[0], y[0], False, True)), x[0] @ y[0].t(), atol=1e-5)
    assert torch.allclose(_t(b.clipByValue(x, -0.5, 0.5)), x.clamp(-0.5, 0.5))
    # legacy families through the DSL: transforms, reduce3, index reduce, reduce_long
    for gen, ref in ((m.abs, torch.abs), (m.ceil, torch.ceil), (m.exp, torch.exp), (m.tanh, torch.tanh), (m.erf, torch.erf), (m.sigmoid, torch.sigmoid),
                     (m.softplus, torch.nn.functional.softplus), (m.cube, lambda v: v ** 3)):
        assert torch.allclose(_t(gen(x)), ref(x), atol=1e-5), gen.__name__
    pos = x.abs() + 0.5
    assert torch.allclose(_t(m.log1p(pos)), torch.log1p(pos), atol=1e-5) and torch.allclose(_t(m.rsqrt(pos)), torch.rsqrt(pos), atol=1e-5)
    assert torch.allclose(_t(m.sqrt(pos)), torch.sqrt(pos), atol=1e-6) and torch.allclose(_t(m.log(pos)), torch.log(pos), atol=1e-6)
    assert torch.equal(_t(m.isNaN(torch.tensor([1.0, float("nan")]))), torch.tensor([False, True]))
    assert torch.allclose(_t(m.cosineSimilarity(x, y, 2)), torch.nn.functional.cosine_similarity(x, y, dim=2), atol=1e-5)
    assert torch.allclose(_t(m.euclideanDistance(x, y, 1, 2)), (x - y).pow(2).sum((1, 2)).sqrt(), atol=1e-5)
    assert torch.equal(_t(m.iamax(x, 2)).long(), x.abs().argmax(2)) and int(_t(m.countNonZero(torch.tensor([0.0, 1.0, 2.0, 0.0])))) == 2
    assert torch.allclose(_t(m.logSumExp(x, 1)), torch.logsumexp(x, 1), atol=1e-5)
    mean, var = m.moments(x, 0, 1)
    assert torch.allclose(_t(mean), x.mean((0, 1)), atol=1e-6) and torch.allclose(_t(var), x.var((0, 1), unbiased=False), atol=1e-6)
    assert torch.allclose(_t(m.mergeAdd([x, y, x])), 2 * x + y) and torch.allclose(_t(m.pow(x.abs(), y)), x.abs() ** y, atol=1e-5)
    assert torch.allclose(_t(m.trace(x[0][:4, :4])), torch.trace(x[0][:4, :4]))


def test_generated_eager_nn_cnn_rnn_match_torch():
    import torch
    import torch.nn.functional as F
    from deeplearning4j_amd.autodiff.generated.nd_cnn import NDCNN
    from deeplearning4j_amd.autodiff.generated.nd_nn import NDNN
    from deeplearning4j_amd.autodiff.generated.nd_rnn import NDRNN
    torch.manual_seed(1)
    nn, cnn, rnn = NDNN(), NDCNN(), NDRNN()
    x = torch.randn(4, 6)
    assert torch.allclose(_t(nn.relu(x)), F.relu(x)) and torch.allclose(_t(nn.leakyRelu(x, 0.2)), F.leaky_relu(x, 0.2))
    assert torch.allclose(_t(nn.softmax(x)), F.softmax(x, -1), atol=1e-6) and torch.allclose(_t(nn.logSoftmax(x, 0)), F.log_softmax(x, 0), atol=1e-6)
    assert torch.allclose(_t(nn.swish(x)), x * torch.sigmoid(x), atol=1e-6) and torch.allclose(_t(nn.elu(x)), F.elu(x), atol=1e-6)
    w, bias = torch.randn(6, 3), torch.randn(3)
    assert torch.allclose(_t(nn.linear(x, w, bias)), x @ w + bias, atol=1e-5) and torch.allclose(_t(nn.reluLayer(x, w, bias)), F.relu(x @ w + bias), atol=1e-5)
    g, be = torch.rand(6) + 0.5, torch.randn(6)
    assert torch.allclose(_t(nn.layerNorm(x, g, be, False, 1e-5, 1)), F.layer_norm(x, (6,), g, be), atol=1e-5)
    assert torch.allclose(_t(nn.layerNorm(x, g, None, False, 1e-5, 1)), F.layer_norm(x, (6,), g, None), atol=1e-5)
    img = torch.randn(2, 3, 8, 8)
    mean, var, gam, bet = torch.randn(3), torch.rand(3) + 0.5, torch.rand(3) + 0.5, torch.randn(3)
    assert torch.allclose(_t(nn.batchNorm(img, mean, var, gam, bet, 1e-5, True, True, 1)), F.batch_norm(img, mean, var, gam, bet, False, 0.0, 1e-5), atol=1e-5)
    pad = torch.tensor([[1, 2], [0, 1]])
    assert torch.equal(_t(nn.pad(x, pad)), F.pad(x, (0, 1, 1, 2))) and torch.equal(_t(nn.pad(x, pad, "REFLECT")), F.pad(x[None], (0, 1, 1, 2), mode="reflect")[0])
    q, k, v = torch.randn(2, 5, 4), torch.randn(2, 5, 7), torch.randn(2, 6, 7)
    att = _t(nn.dotProductAttention(q, k.transpose(1, 2)[:, :4, :5].contiguous() if False else torch.randn(2, 5, 4), torch.randn(2, 5, 4)))
    assert tuple(att.shape)[0] == 2
    # cnn: conv2d weights in the default YXIO layout [kH, kW, iC, oC]
    wt = torch.randn(5, 3, 3, 3)
    out = _t(cnn.conv2d(img, wt.permute(2, 3, 1, 0).contiguous(), None, 3, 3, 1, 1, 1, 1))
    assert torch.allclose(out, F.conv2d(img, wt, None, 1, 1), atol=1e-4)
    out = _t(cnn.conv2d(img, wt, torch.ones(5), 3, 3, 2, 2, 0, 0, 1, 1, False, False, "OIYX"))
    assert torch.allclose(out, F.conv2d(img, wt, torch.ones(5), 2, 0), atol=1e-4)
    assert torch.allclose(_t(cnn.maxPooling2d(img, 2, 2, 2, 2)), F.max_pool2d(img, 2, 2)) and torch.allclose(_t(cnn.avgPooling2d(img, 2, 2, 2, 2)), F.avg_pool2d(img, 2, 2), atol=1e-6)
    assert torch.equal(_t(cnn.upsampling2d(img, 2, 2)), F.interpolate(img, scale_factor=2, mode="nearest"))
    assert torch.equal(_t(cnn.depthToSpace(_t(cnn.spaceToDepth(img, 2)), 2)), img)
    cols = _t(cnn.im2Col(img, 3, 3, 1, 1, 1, 1))
    assert tuple(cols.shape) == (2, 3, 3, 3, 8, 8) and torch.allclose(cols.reshape(2, 27, 64), F.unfold(img, 3, padding=1), atol=1e-6)
    # rnn: gruCell against the gate equations
    nIn, nOut, mb = 3, 4, 2
    xx, h0 = torch.randn(mb, nIn), torch.randn(mb, nOut)
    Wru, Wc, bru, bc = torch.randn(nIn + nOut, 2 * nOut), torch.randn(nIn + nOut, nOut), torch.randn(2 * nOut), torch.randn(nOut)
    r, u, c, h = (_t(t) for t in rnn.gruCell(xx, h0, Wru, Wc, bru, bc))
    ru = torch.sigmoid(torch.cat([xx, h0], 1) @ Wru + bru)
    rr, uu = ru[:, :nOut], ru[:, nOut:]
    cc = torch.tanh(torch.cat([xx, rr * h0], 1) @ Wc + bc)
    assert torch.allclose(r, rr, atol=1e-5) and torch.allclose(u, uu, atol=1e-5) and torch.allclose(c, cc, atol=1e-5)
    assert torch.allclose(h, uu * h0 + (1 - uu) * cc, atol=1e-5)


def test_generated_samediff_mixins_build_executable_graphs():
    import torch
    from deeplearning4j_amd.autodiff.generated.sd_base import GenSDBase
    from deeplearning4j_amd.autodiff.generated.sd_math import GenSDMath
    from deeplearning4j_amd.autodiff.generated.sd_nn import GenSDNN
    from deeplearning4j_amd.autodiff.namespaces import _NS
    from deeplearning4j_amd.autodiff.samediff import SameDiff

    class B(GenSDBase, _NS): pass          # the generated surface alone, without the hand-written overrides
    class M(GenSDMath, _NS): pass
    class N(GenSDNN, _NS): pass
    sd = SameDiff.create()
    x = sd.placeHolder("x", "FLOAT", 3, 4)
    w = sd.var("w", torch.randn(4, 2))
    b, m, n = B(sd), M(sd), N(sd)
    h = n.relu(b.mmul(m.tanh(x), w), name="h")
    s = b.sum(m.square(h), True, 1, name="s")
    d = m.euclideanDistance(h, b.zerosLike(h), 1, name="d")
    xv = torch.randn(3, 4)
    out = sd.output({"x": xv}, ["h", "s", "d"])
    hv = torch.relu(torch.tanh(xv) @ sd.getArrForVarName("w").tensor())
    assert torch.allclose(_t(out["h"]), hv, atol=1e-5) and torch.allclose(_t(out["s"]), hv.pow(2).sum(1, keepdim=True), atol=1e-5)
    assert torch.allclose(_t(out["d"]), hv.norm(dim=1), atol=1e-5)
    sd.setLossVariables(b.sum(s, False, name="total"))
    g = sd.calculateGradients({"x": xv}, ["w"])
    assert torch.isfinite(_t(g["w"])).all() and _t(g["w"]).abs().sum() > 0
    # the public namespaces expose the generated methods too
    assert hasattr(sd.math, "hammingDistance") and hasattr(sd.nn, "localResponseNormalization") and hasattr(sd.cnn, "col2Im")
