"""Evaluation family + normalizers (mirrors nd4j EvalTest / RegressionEvalTest / ROCTest / NormalizerTests:
hand-computed expectations, merge() == single-pass, serializer round trips)."""
import io
import math

import pytest
import torch

from deeplearning4j_amd.datasets.dataset import DataSet, MultiDataSet
from deeplearning4j_amd.datasets.iterators import ListDataSetIterator
from deeplearning4j_amd.datasets import normalizers as N
from deeplearning4j_amd.eval.evaluation import (Evaluation, EvaluationBinary, EvaluationCalibration, ROC, ROCBinary,
                                                ROCMultiClass, RegressionEvaluation)


def test_evaluation_counts_and_metrics():
    # actual: 0 0 1 1 2 2 ; predicted: 0 1 1 1 2 0
    y = torch.eye(3)[[0, 0, 1, 1, 2, 2]]
    p = torch.eye(3)[[0, 1, 1, 1, 2, 0]] * 0.8 + 0.1
    e = Evaluation()
    e.eval(y, p)
    assert e.accuracy() == pytest.approx(4 / 6)
    assert e.truePositives(1) == 2 and e.falsePositives(1) == 1 and e.falseNegatives(0) == 1
    assert e.precision(1) == pytest.approx(2 / 3)
    assert e.recall(0) == pytest.approx(0.5)
    assert e.precision() == pytest.approx((0.5 + 2 / 3 + 1.0) / 3)
    assert e.recall() == pytest.approx((0.5 + 1.0 + 0.5) / 3)
    assert e.getConfusionMatrix().getCount(2, 0) == 1
    assert "Accuracy" in e.stats()


def test_evaluation_merge_topn_timeseries_mask():
    g = torch.Generator().manual_seed(1)
    y = torch.eye(5)[torch.randint(0, 5, (40,), generator=g)]
    p = torch.softmax(torch.randn(40, 5, generator=g), 1)
    full = Evaluation(topN=3)
    full.eval(y, p)
    a, b = Evaluation(topN=3), Evaluation(topN=3)
    a.eval(y[:17], p[:17])
    b.eval(y[17:], p[17:])
    a.merge(b)
    assert a.accuracy() == full.accuracy() and a.topNAccuracy() == full.topNAccuracy()
    assert full.topNAccuracy() >= full.accuracy()
    # time series [mb, n, T] with mask == flattened, masked rows dropped
    yt = y.reshape(8, 5, 5).permute(0, 2, 1).contiguous()        # [8, n=5, T=5]
    pt = p.reshape(8, 5, 5).permute(0, 2, 1).contiguous()
    mask = torch.ones(8, 5)
    mask[:, -1] = 0
    ts = Evaluation()
    ts.evalTimeSeries(yt, pt, mask)
    keep = mask.reshape(-1) > 0
    flat = Evaluation()
    flat.eval(y[keep], p[keep])
    assert ts.accuracy() == flat.accuracy() and ts.getNumRowCounter() == 32


def test_evaluation_binary_and_regression():
    y = torch.tensor([[1, 0], [1, 1], [0, 1], [0, 0.0]])
    p = torch.tensor([[0.9, 0.2], [0.4, 0.8], [0.6, 0.7], [0.1, 0.6]])
    eb = EvaluationBinary()
    eb.eval(y, p)
    assert eb.truePositives(0) == 1 and eb.falseNegatives(0) == 1 and eb.falsePositives(0) == 1 and eb.trueNegatives(0) == 1
    assert eb.accuracy(1) == pytest.approx(0.75)
    r = RegressionEvaluation()
    yy = torch.tensor([[1.0, 2.0], [2.0, 4.0], [3.0, 6.0]])
    pp = torch.tensor([[1.5, 2.0], [2.0, 3.0], [2.5, 6.5]])
    r.eval(yy, pp)
    assert r.meanSquaredError(0) == pytest.approx((0.25 + 0 + 0.25) / 3)
    assert r.meanAbsoluteError(1) == pytest.approx((0 + 1 + 0.5) / 3)
    assert r.rootMeanSquaredError(0) == pytest.approx(math.sqrt(0.5 / 3))
    # vs direct formulas
    sstot = ((yy[:, 1] - yy[:, 1].mean()) ** 2).sum()
    assert r.rSquared(1) == pytest.approx(float(1 - ((pp[:, 1] - yy[:, 1]) ** 2).sum() / sstot))
    pc = torch.corrcoef(torch.stack([yy[:, 0], pp[:, 0]]))[0, 1]
    assert r.pearsonCorrelation(0) == pytest.approx(float(pc), abs=1e-6)
    r2 = RegressionEvaluation()
    r2.eval(yy[:1], pp[:1])
    r3 = RegressionEvaluation()
    r3.eval(yy[1:], pp[1:])
    r2.merge(r3)
    assert r2.averageMeanSquaredError() == pytest.approx(r.averageMeanSquaredError())


def _auc_ref(y, p):
    pos, neg = p[y > 0.5], p[y <= 0.5]
    gt = (pos.unsqueeze(1) > neg.unsqueeze(0)).double().sum()
    eq = (pos.unsqueeze(1) == neg.unsqueeze(0)).double().sum()
    return float((gt + 0.5 * eq) / (len(pos) * len(neg)))


def test_roc_exact_thresholded_multiclass():
    g = torch.Generator().manual_seed(3)
    y = (torch.rand(300, generator=g) > 0.6).float()
    p = (0.3 * y + 0.7 * torch.rand(300, generator=g)).clamp(0, 1)
    roc = ROC(0)
    roc.eval(y.unsqueeze(1), p.unsqueeze(1))
    assert roc.calculateAUC() == pytest.approx(_auc_ref(y, p), abs=1e-9)
    assert 0.5 < roc.calculateAUCPR() <= 1.0
    rt = ROC(200)
    rt.eval(torch.stack([1 - y, y], 1), torch.stack([1 - p, p], 1))
    assert rt.calculateAUC() == pytest.approx(roc.calculateAUC(), abs=5e-3)
    a, b = ROC(0), ROC(0)
    a.eval(y[:100].unsqueeze(1), p[:100].unsqueeze(1))
    b.eval(y[100:].unsqueeze(1), p[100:].unsqueeze(1))
    assert a.merge(b).calculateAUC() == pytest.approx(roc.calculateAUC(), abs=1e-12)
    # perfect separation
    pr = ROC(0)
    pr.eval(torch.tensor([[0.], [0.], [1.], [1.]]), torch.tensor([[0.1], [0.2], [0.8], [0.9]]))
    assert pr.calculateAUC() == pytest.approx(1.0) and pr.calculateAUCPR() == pytest.approx(1.0)
    ym = torch.eye(3)[torch.randint(0, 3, (120,), generator=g)]
    pm = torch.softmax(torch.randn(120, 3, generator=g) + 2 * ym, 1)
    rm = ROCMultiClass(0)
    rm.eval(ym, pm)
    for c in range(3):
        assert rm.calculateAUC(c) == pytest.approx(_auc_ref(ym[:, c], pm[:, c]), abs=1e-9)
    rb = ROCBinary(0)
    rb.eval(ym, pm)
    assert rb.calculateAverageAUC() == pytest.approx(rm.calculateAverageAUC())


def test_calibration():
    g = torch.Generator().manual_seed(5)
    p = torch.rand(5000, 1, generator=g)
    y = (torch.rand(5000, 1, generator=g) < p).float()         # perfectly calibrated
    p2, y2 = torch.cat([1 - p, p], 1), torch.cat([1 - y, y], 1)
    ec = EvaluationCalibration(10, 20)
    ec.eval(y2, p2)
    meanPred, fracPos = ec.getReliabilityDiagram(1)
    assert (meanPred - fracPos).abs().max() < 0.08
    assert ec.expectedCalibrationError(1) < 0.04
    assert int(ec.getProbabilityHistogramAllClasses().sum()) == 10000


def _ds(n=50, seed=0):
    g = torch.Generator().manual_seed(seed)
    return DataSet(torch.randn(n, 4, generator=g) * torch.tensor([1.0, 5.0, 0.1, 2.0]) + torch.tensor([0.0, 3.0, -1.0, 10.0]),
                   torch.randn(n, 2, generator=g) * 3 + 1)


def test_standardize_fit_iterator_equals_full_and_roundtrip():
    ds = _ds()
    it = ListDataSetIterator(ds.batchBy(7), 7)
    n = N.NormalizerStandardize().fitLabel(True)
    n.fit(it)
    assert torch.allclose(n.getMean(), ds.features.mean(0), atol=1e-5)
    assert torch.allclose(n.getStd(), ds.features.std(0, unbiased=False), atol=1e-5)
    c = ds.copy()
    n.transform(c)
    assert c.features.mean(0).abs().max() < 1e-5 and (c.features.std(0, unbiased=False) - 1).abs().max() < 1e-4
    assert c.labels.mean(0).abs().max() < 1e-5
    n.revert(c)
    assert torch.allclose(c.features, ds.features, atol=1e-4) and torch.allclose(c.labels, ds.labels, atol=1e-4)
    b = io.BytesIO()
    N.NormalizerSerializer.getDefault().write(n, b)
    r = N.NormalizerSerializer.getDefault().restore(io.BytesIO(b.getvalue()))
    assert isinstance(r, N.NormalizerStandardize) and r.isFitLabel()
    assert torch.equal(r.getMean(), n.getMean()) and torch.equal(r.getLabelStd(), n.getLabelStd())


def test_minmax_image_vgg_and_serializer():
    ds = _ds()
    n = N.NormalizerMinMaxScaler(-1, 1)
    n.fit(ds)
    c = ds.copy()
    n.transform(c)
    assert torch.allclose(c.features.amin(0), torch.full((4,), -1.0), atol=1e-5)
    assert torch.allclose(c.features.amax(0), torch.full((4,), 1.0), atol=1e-5)
    n.revert(c)
    assert torch.allclose(c.features, ds.features, atol=1e-4)
    S = N.NormalizerSerializer.getDefault()
    for norm in (n, N.ImagePreProcessingScaler(-1, 1), N.VGG16ImagePreProcessor()):
        b = io.BytesIO()
        S.write(norm, b)
        r = S.restore(io.BytesIO(b.getvalue()))
        assert type(r) is type(norm)
    img = DataSet(torch.full((2, 3, 4, 4), 255.0), torch.zeros(2, 1))
    N.ImagePreProcessingScaler(0, 1).transform(img)
    assert torch.allclose(img.features, torch.ones(2, 3, 4, 4))
    ts = DataSet(torch.randn(6, 3, 9) * 4 + 2, torch.zeros(6, 1))
    ns = N.NormalizerStandardize().fit(ts)
    ns.transform(ts)
    assert ts.features.permute(0, 2, 1).reshape(-1, 3).mean(0).abs().max() < 1e-5


def test_multi_normalizers():
    g = torch.Generator().manual_seed(2)
    mds = MultiDataSet([torch.randn(30, 3, generator=g) * 4 + 1, torch.randn(30, 2, generator=g) - 7],
                       [torch.randn(30, 2, generator=g) * 2])
    n = N.MultiNormalizerStandardize().fitLabel(True)
    n.fit(mds)
    b = io.BytesIO()
    N.NormalizerSerializer.getDefault().write(n, b)
    r = N.NormalizerSerializer.getDefault().restore(io.BytesIO(b.getvalue()))
    assert r.numInputs() == 2 and r.numOutputs() == 1 and torch.equal(r.getFeatureMean(1), n.getFeatureMean(1))
    orig = [f.clone() for f in mds.features]
    n.transform(mds)
    assert mds.features[1].mean(0).abs().max() < 1e-5
    n.revert(mds)
    assert torch.allclose(mds.features[0], orig[0], atol=1e-4)
    mm = N.MultiNormalizerMinMaxScaler().fit(mds)
    mm.transform(mds)
    assert float(mds.features[0].amin()) == pytest.approx(0.0, abs=1e-6)
    h = N.MultiNormalizerHybrid().standardizeInput(0).minMaxScaleInput(1).minMaxScaleAllOutputs(-1, 1)
    m2 = MultiDataSet([o.clone() for o in orig], [mds.labels[0].clone()])
    h.fit(m2)
    h.transform(m2)
    assert m2.features[0].mean(0).abs().max() < 1e-5
    assert float(m2.features[1].amax()) == pytest.approx(1.0, abs=1e-6)
    assert float(m2.labels[0].amin()) == pytest.approx(-1.0, abs=1e-6)


def test_model_serializer_with_normalizer(tmp_path):
    from deeplearning4j_amd.nn.conf.configuration import NeuralNetConfiguration
    from deeplearning4j_amd.nn.conf.layers import DenseLayer, OutputLayer
    from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
    from deeplearning4j_amd.utils.model_serializer import ModelSerializer
    conf = (NeuralNetConfiguration.Builder().seed(1).list()
            .layer(DenseLayer.Builder().nIn(4).nOut(5).activation("tanh").build())
            .layer(OutputLayer.Builder().nIn(5).nOut(2).activation("softmax").lossFunction("MCXENT").build()).build())
    net = MultiLayerNetwork(conf)
    net.init()
    n = N.NormalizerStandardize().fit(_ds())
    p = tmp_path / "m.zip"
    ModelSerializer.writeModel(net, str(p), True, n)
    r = ModelSerializer.restoreNormalizerFromFile(str(p))
    assert torch.equal(r.getMean(), n.getMean())
    net2 = ModelSerializer.restoreMultiLayerNetwork(str(p))
    assert torch.equal(net2.params(), net.params())
    ev = net2.evaluate(ListDataSetIterator([DataSet(torch.randn(8, 4), torch.eye(2)[torch.randint(0, 2, (8,))])], 8))
    assert 0.0 <= ev.accuracy() <= 1.0
