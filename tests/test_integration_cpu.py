"""IntegrationTestRunner-style phases on the CPU: predictions, gradients, training curve, parameters after N steps, evaluation,
serialization round trip, ParallelInference agreement, overfitting, layer clearance and iteration counters (see integration_cases.py)."""
import io

import pytest
import torch

from deeplearning4j_amd.datasets.dataset import DataSet
from deeplearning4j_amd.datasets.iterators import ListDataSetIterator
from deeplearning4j_amd.nn.multilayer import MultiLayerNetwork
from deeplearning4j_amd.parallel.parallel_inference import ParallelInference, InferenceMode
from integration_cases import CASES


def _curve(net, x, y, steps):
    out = []
    for _ in range(steps):
        net.fit(x, y)
        out.append(float(net.score()))
    return out


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_integration_phases(case, tmp_path):
    x, y = case.data()
    a, b = case.net("cpu"), case.net("cpu")
    # predictions + gradients of two independently built networks (same seed) agree exactly
    assert torch.equal(a.output(x), b.output(x))
    a.computeGradientAndScore(x, y)
    b.computeGradientAndScore(x, y)
    assert torch.equal(a.gradient().reshape(-1) if hasattr(a.gradient(), "reshape") else a._grads, b.gradient().reshape(-1) if hasattr(b.gradient(), "reshape") else b._grads)
    # training curve and parameters after N steps are reproducible (MAX_REL_ERROR_SCORES = 1e-4 in the reference runner)
    ca, cb = _curve(a, x, y, 25), _curve(b, x, y, 25)
    for u, v in zip(ca, cb):
        assert abs(u - v) <= case.atol_curve * max(1.0, abs(u))
    assert torch.allclose(a.params(), b.params(), atol=1e-6)
    assert ca[-1] < ca[0]
    # iteration / epoch counters (validateLayerIterCounts) and layer clearance (checkLayerClearance)
    assert a.getIterationCount() == 25
    a.clear()
    for l in a.layers:
        assert l.input is None and l.maskArray is None
    # serialization: the restored network gives the same predictions, continues training identically (updater state saved)
    path = tmp_path / (case.name + ".zip")
    a.save(str(path), True)
    r = MultiLayerNetwork.load(str(path), True, device="cpu")
    assert torch.equal(r.output(x), a.output(x))
    assert r.getIterationCount() == a.getIterationCount()
    a.fit(x, y)
    r.fit(x, y)
    assert torch.allclose(r.params(), a.params(), atol=1e-6)
    # evaluation through an iterator
    it = ListDataSetIterator(DataSet(x, y).batchBy(16), 16)
    ev = a.evaluate(it)
    assert 0.0 <= ev.accuracy() <= 1.0
    # ParallelInference phase: same inputs through the batched server
    pi = ParallelInference(a, workers=2, inferenceMode=InferenceMode.BATCHED, batchLimit=8)
    try:
        for k in range(0, 24, 6):
            got = pi.output(x[k:k + 6])
            got = got[0] if isinstance(got, (list, tuple)) else got
            assert torch.allclose(got, a.output(x[k:k + 6]), atol=1e-6)
    finally:
        pi.shutdown()
    # overfitting phase: a handful of examples is memorised
    o = case.net("cpu")
    xs, ys = x[:8], y[:8]
    for _ in range(150):
        o.fit(xs, ys)
    pred, tgt = o.output(xs), ys
    acc = float((pred.argmax(1) == tgt.argmax(1)).float().mean())
    assert acc >= 0.9, acc
