"""The same registry sweep on the GPU: every case's result on cuda tensors against the independent reference (so every @platform HIP
route that a case reaches is numerically checked against NumPy / fp64 math), plus the accounting that shows which ops ran on their
hand-written kernel and that nothing fell back unseen."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _kernels_not_cost_model():
    """These tests check the hand-written kernels themselves: the measured cost model (which would route some shapes to ATen) is off."""
    from deeplearning4j_amd.nd4j.ops import cost_model
    old, cost_model.ENABLED = cost_model.ENABLED, False
    yield
    cost_model.ENABLED = old


def test_registry_sweep_on_gpu_and_helper_accounting():
    from deeplearning4j_amd.autodiff.validation import sweep, summarize
    from deeplearning4j_amd.nd4j.ops.custom import OpExecutioner, _PLATFORM
    OpExecutioner.resetHelperStatistics()
    rep = sweep("cuda")
    s = summarize(rep)
    failed = {n: e["errors"] for n, e in rep.items() if e["status"] == "failed"}
    assert not failed, failed
    assert s["covered_pct"] >= 99.9 and s["grad_failed"] == 0 and s["reference"] >= 480, s
    st = OpExecutioner.helperStatistics()
    ran = set(st["hit"])
    assert len(ran) >= 40, sorted(ran)                                  # hand-written routes really executed during the sweep
    # every op with a platform route that the sweep executed is accounted for: it ran its helper, or its rejection was counted
    seen = ran | {k[0] for k in st["fail"]}
    executed = {n for n, e in rep.items() if e["cases"] and n in _PLATFORM}
    assert executed <= seen, sorted(executed - seen)
