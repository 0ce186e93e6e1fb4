"""Native op-execution ABI on DEVICE buffers: the cases of abi_cases.py through execCustomOp2 with no fallback registered
(so a pass cannot come from the Python op library), and a whole LeNet training step - forward, backward, Adam - driven only by
ctypes calls into libnd4jrocm_rt.so on device pointers, compared with the Python MultiLayerNetwork."""
import pytest
import torch

from abi_cases import cases, lenet_step_check
from test_native_abi_ops_cpu import check, reference
from deeplearning4j_amd.nd4j import native_ops as N

pytestmark = pytest.mark.gpu
CASES = cases()


@pytest.fixture(autouse=True)
def _no_fallback():
    N.uninstall_fallback()
    yield


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_native_op_on_device_buffers(case):
    ins = [t.cuda() for t in case["ins"]]
    got = N.exec_custom_op(case["name"], ins, iArgs=case["i"], tArgs=case["t"], bArgs=case["b"], dArgs=case["d"])
    torch.cuda.synchronize()
    assert all(g.is_cuda for g in got)
    check(case, got, reference(case))


def test_lenet_training_step_through_the_c_abi_only():
    lenet_step_check("cuda")
