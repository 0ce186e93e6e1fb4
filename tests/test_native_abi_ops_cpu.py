"""Native op-execution ABI: every op of the native table (rt_opctx.cpp + rt_opctx_nn.inc) runs through execCustomOp2 on HOST
arrays with no fallback registered and reproduces the Python op library; shapes come from the native shape functions
(calculateOutputShapes2). The same cases run on device buffers in tests/test_native_abi_ops_gpu.py."""
import pytest
import torch

from abi_cases import cases, lenet_step_check
from deeplearning4j_amd.nd4j import native_ops as N
from deeplearning4j_amd.nd4j.ops import custom

_DT = {9: torch.int32, 5: torch.float32, 17: torch.bfloat16, 10: torch.int64, 6: torch.float64}
CASES = cases()


def reference(case, device="cpu"):
    ins = [t.to(device) for t in case["ins"]]
    custom._load_impls()      # the generic implementations themselves: no descriptor validation, no platform helpers
    res = custom._IMPL[case["name"]](ins, case["i"], case["t"], case["b"], [_DT[v] for v in case["d"]])
    return list(res) if isinstance(res, (list, tuple)) else [res]


def check(case, got, want):
    assert len(got) >= 1 and len(got) <= len(want) or len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        w = w.to(g.device)
        assert g.numel() == w.numel(), "%s output %d: %s vs %s" % (case["id"], k, tuple(g.shape), tuple(w.shape))
        if g.dtype == torch.bool or not g.dtype.is_floating_point:
            assert torch.equal(g.reshape(-1).to(torch.int64), w.reshape(-1).to(torch.int64)), "%s output %d" % (case["id"], k)
        else:
            tol = 1e-2 if g.dtype == torch.bfloat16 else case["tol"]
            torch.testing.assert_close(g.reshape(-1).float(), w.reshape(-1).float(), rtol=tol, atol=tol, msg=lambda m: "%s output %d: %s" % (case["id"], k, m))


@pytest.fixture(autouse=True)
def _no_fallback():
    N.uninstall_fallback()
    yield


def test_native_table_size_and_listing():
    names = N.native_op_names()
    assert len(names) >= 60 and len(names) == N.lib().nativeOpCount()
    for must in ("conv2d", "conv2d_bp", "batchnorm", "batchnorm_bp", "maxpool2d", "maxpool2d_bp", "avgpool2d", "avgpool2d_bp",
                 "softmax_cross_entropy_loss", "softmax_cross_entropy_loss_grad", "layer_norm", "layer_norm_bp", "matmul_bp", "gather",
                 "scatter_add", "concat", "cast", "adam_updater", "sgd_updater", "nesterovs_updater", "rms_prop_updater", "ada_grad_updater",
                 "ada_delta_updater", "ada_max_updater", "nadam_updater", "ams_grad_updater", "adabelief_updater"):
        assert must in names, must
    listing = N.list_custom_ops()
    assert listing["conv2d"] == "native" and listing["lstmLayer"] == "fallback"
    assert sum(v == "native" for v in listing.values()) == len(names)
    assert set(c["name"] for c in CASES) >= set(names), sorted(set(names) - set(c["name"] for c in CASES))


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_native_op_on_host_arrays(case):
    assert N.lib().nativeOpImplemented(N.lib().ops_hash(case["name"].encode()))
    got = N.exec_custom_op(case["name"], case["ins"], iArgs=case["i"], tArgs=case["t"], bArgs=case["b"], dArgs=case["d"])
    check(case, got, reference(case))


def test_error_paths_report_instead_of_crashing():
    x = torch.randn(2, 3, 4, 4)
    with pytest.raises(RuntimeError, match="weights' input channels"):
        N.exec_custom_op("conv2d", [x, torch.randn(4, 5, 3, 3)], [torch.empty(2, 4, 2, 2)], iArgs=[3, 3, 1, 1, 0, 0, 1, 1, 0, 0, 1])
    with pytest.raises(RuntimeError, match="output shape"):
        N.exec_custom_op("maxpool2d", [x], [torch.empty(2, 3, 3, 3)], iArgs=[2, 2, 2, 2, 0, 0, 1, 1, 0, 0, 0])
    with pytest.raises(RuntimeError, match="permutation"):
        N.exec_custom_op("permute", [x], [torch.empty(4, 4, 3, 2)], iArgs=[3, 3, 1, 0])
    with pytest.raises(RuntimeError, match="share one length"):
        N.exec_custom_op("adam_updater", [torch.randn(8), torch.randn(8), torch.randn(7)], [torch.empty(8), torch.empty(8), torch.empty(8)])
    # out-of-range gather indices are skipped, never read
    out = N.exec_custom_op("gather", [torch.arange(6.0).reshape(3, 2), torch.tensor([0, 7])], [torch.full((2, 2), -1.0)], iArgs=[0])[0]
    assert out[0].tolist() == [0.0, 1.0] and out[1].tolist() == [-1.0, -1.0]


def test_lenet_training_step_through_the_c_abi_only_host_buffers():
    lenet_step_check("cpu", n=8)
