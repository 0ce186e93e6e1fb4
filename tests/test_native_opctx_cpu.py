"""The native op-execution ABI driven from ctypes the way the reference's JavaInteropTests drive it (libnd4j/tests_cpu/layers_tests/
JavaInteropTests.cpp: build shapeInfo, fill an OpaqueContext, execCustomOp2 by op hash, calculateOutputShapes2) — host arrays here,
device arrays in tests/test_native_opctx_gpu.py. References are plain PyTorch / NumPy math."""
import ctypes as C

import pytest
import torch

from deeplearning4j_amd.nd4j import native_ops as N


def test_native_table_and_shape_functions():
    rt = N.lib()
    assert rt.nativeOpCount() >= 15
    for name in ("add", "matmul", "softmax", "relu", "biasadd", "reduce_mean"):
        assert rt.nativeOpImplemented(C.c_int64(rt.ops_hash(name.encode()))) == 1
    assert rt.nativeOpImplemented(C.c_int64(rt.ops_hash(b"cumsum"))) == 0
    a, b = torch.zeros(3, 1, 4), torch.zeros(5, 1)
    assert N.calculate_output_shapes("add", [a, b]) == [([3, 5, 4], torch.float32)]
    assert N.calculate_output_shapes("matmul", [torch.zeros(3, 4, dtype=torch.float64), torch.zeros(5, 4, dtype=torch.float64)], iArgs=[0, 1, 0]) == [([3, 5], torch.float64)]
    assert N.calculate_output_shapes("reduce_sum", [torch.zeros(2, 3, 4)], iArgs=[1], bArgs=[True]) == [([2, 1, 4], torch.float32)]
    assert N.calculate_output_shapes("reduce_sum", [torch.zeros(2, 3, 4)], iArgs=[0, 2]) == [([3], torch.float32)]
    assert N.calculate_output_shapes("cumsum", [a]) is None


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_exec_custom_op2_host_arrays(dt):
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(3, 1, 4, generator=g).to(dt), torch.randn(5, 1, generator=g).to(dt)
    assert torch.equal(N.exec_custom_op("add", [a, b])[0], a + b)
    assert torch.equal(N.exec_custom_op("multiply", [a, b])[0], a * b)
    assert torch.equal(N.exec_custom_op("maximum", [a, a.flip(0)])[0], torch.maximum(a, a.flip(0)))
    x = torch.randn(6, 7, generator=g).to(dt)
    assert torch.allclose(N.exec_custom_op("relu", [x], tArgs=[0.0])[0], x.clamp_min(0))
    assert torch.allclose(N.exec_custom_op("sigmoid", [x])[0], torch.sigmoid(x), atol=1e-6)
    assert torch.allclose(N.exec_custom_op("tanh", [x])[0], torch.tanh(x), atol=1e-6)
    assert torch.allclose(N.exec_custom_op("softmax", [x])[0], torch.softmax(x, -1), atol=1e-6)
    w = torch.randn(5, 7, generator=g).to(dt)
    assert torch.allclose(N.exec_custom_op("matmul", [x, w], iArgs=[0, 1, 0])[0], x @ w.t(), atol=1e-5)
    assert torch.allclose(N.exec_custom_op("matmul", [x.t().contiguous(), w.t().contiguous()], iArgs=[1, 0, 0], tArgs=[2.0, 0.0])[0], 2 * x @ w.t(), atol=1e-5)
    bias = torch.randn(7, generator=g).to(dt)
    assert torch.allclose(N.exec_custom_op("biasadd", [x, bias])[0], x + bias)
    img = torch.randn(2, 3, 4, 4, generator=g).to(dt)
    assert torch.allclose(N.exec_custom_op("biasadd", [img, bias[:3].contiguous()], bArgs=[True])[0], img + bias[:3].view(1, 3, 1, 1))
    assert torch.allclose(N.exec_custom_op("reduce_mean", [img], iArgs=[0, 2, 3])[0], img.mean((0, 2, 3)), atol=1e-6)
    assert torch.allclose(N.exec_custom_op("reduce_sum", [img], iArgs=[1], bArgs=[True])[0], img.sum(1, keepdim=True), atol=1e-5)


def test_errors_cross_the_abi_as_status_codes():
    with pytest.raises(RuntimeError, match="broadcast"):
        N.exec_custom_op("add", [torch.zeros(3, 4), torch.zeros(5)], outputs=[torch.zeros(3, 4)])
    N.uninstall_fallback()
    with pytest.raises(RuntimeError, match="not implemented natively"):
        N.exec_custom_op("cumsum", [torch.zeros(3)], outputs=[torch.zeros(3)])


def test_fallback_serves_the_rest_of_the_op_library_through_the_same_entry_point():
    N.install_python_fallback()
    try:
        x = torch.arange(12.0).reshape(3, 4)
        out = N.exec_custom_op("cumsum", [x], outputs=[torch.empty(3, 4)], iArgs=[0, 0, 1])[0]
        assert torch.equal(out, x.cumsum(1))
        # a form the native matmul executor declines (transposed result) is routed to the fallback transparently
        a, b = torch.randn(3, 4), torch.randn(4, 5)
        out = N.exec_custom_op("matmul", [a, b], outputs=[torch.empty(5, 3)], iArgs=[0, 0, 1])[0]
        assert torch.allclose(out, (a @ b).t(), atol=1e-6)
    finally:
        N.uninstall_fallback()
