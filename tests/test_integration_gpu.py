"""Backend parameterisation (BaseNd4jTestWithBackends analogue): every integration case on the MI355X must reproduce the CPU
backend's predictions, gradients and training curve (fp32), and the GPU HIP-graph path must follow the GPU eager path."""
import pytest
import torch

from integration_cases import CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES, ids=[c.name for c in CASES])
def test_gpu_backend_matches_cpu_backend(case):
    x, y = case.data()
    c, g = case.net("cpu"), case.net("cuda")
    xg, yg = x.cuda(), y.cuda()
    assert torch.allclose(g.params().cpu(), c.params(), atol=0)            # same seed -> same init on both backends
    assert torch.allclose(g.output(xg).cpu(), c.output(x), atol=2e-5, rtol=1e-4)
    c.computeGradientAndScore(x, y)
    g.computeGradientAndScore(xg, yg)
    gc, gg = c._grads.reshape(-1), g._grads.reshape(-1).cpu()
    # conv / BN gradients are sums over thousands of terms in a different order on the two backends: compare at the scale of the vector
    assert float((gg - gc).norm() / gc.norm()) < 2e-3 and float((gg - gc).abs().max()) < 5e-3 * float(gc.abs().max())
    assert abs(float(c.score()) - float(g.score())) < 1e-4 * max(1.0, abs(float(c.score())))
    for k in range(20):
        c.fit(x, y)
        g.fit(xg, yg)
        sc, sg = float(c.score()), float(g.score())
        assert abs(sc - sg) < 5e-3 * max(1.0, abs(sc)), (k, sc, sg)
    # HIP-graph captured training follows eager training on the same backend
    e, h = case.net("cuda"), case.net("cuda")
    h.enableGraphCapture(True)
    for k in range(12):
        e.fit(xg, yg)
        h.fit(xg, yg)
    assert torch.allclose(e.params(), h.params(), atol=1e-4, rtol=1e-3), float((e.params() - h.params()).abs().max())
