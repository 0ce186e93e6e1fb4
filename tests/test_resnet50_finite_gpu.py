"""GPU: the benchmark network at the benchmark's image size stays finite.

The kernels' unit tests run small launches; the tile instantiations a 224 x 224 ResNet-50 selects at batch 64-256 are different
ones. A few real training steps must leave a finite score and finite parameters (bench.py refuses to print a non-finite score)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batch,graph", [(64, False), (128, True)])
def test_resnet50_224_training_steps_stay_finite(batch, graph):
    from deeplearning4j_amd.models.resnet50 import ResNet50
    from deeplearning4j_amd.datasets.iterators import BenchmarkDataSetIterator
    dev = torch.device("cuda", 0)
    net = ResNet50(numClasses=1000, inputShape=(3, 224, 224), computeDataType="BFLOAT16").init(device=dev)
    if graph:
        net.enableGraphCapture(True)
    ds = BenchmarkDataSetIterator([batch, 3, 224, 224], 1000, 1, device=dev, seed=1234).next()
    x, y = ds.getFeatures(), ds.getLabels()
    scores = []
    for _ in range(4):
        net.fit(x, y)
        scores.append(float(net.score()))
    assert all(math.isfinite(s) for s in scores), scores
    bad = []
    for name, p in net.paramTable().items():
        t = p.data if hasattr(p, "data") else p
        t = t.tensor() if hasattr(t, "tensor") else t
        if not bool(torch.isfinite(t.float()).all()):
            bad.append(name)
    assert not bad, bad[:8]
