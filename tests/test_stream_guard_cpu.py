"""ops/streams.py vendor-GEMM rule (reference: the per-device mutex around cuBLAS in helpers/cuda_off/MmulHelper.cu:261): the guard's
interception logic on CPU tensors (the mode is device-agnostic; `device_types` selects what it acts on)."""
import pytest
import torch

from deeplearning4j_amd.ops import streams as S


def test_assert_mode_raises_on_every_gemm_spelling_and_lets_other_ops_through():
    a, b = torch.randn(4, 8), torch.randn(8, 3)
    with S.library_gemm_guard("assert", "test region", device_types=("cpu",)):
        assert torch.allclose(a + 1 - 1, a, atol=1e-6)                   # elementwise work is untouched
        assert torch.relu(a).shape == a.shape
        for call in (lambda: torch.mm(a, b), lambda: a @ b, lambda: a.matmul(b), lambda: torch.addmm(torch.zeros(3), a, b),
                     lambda: torch.nn.functional.linear(a, b.t()), lambda: torch.einsum("ik,kj->ij", a, b),
                     lambda: torch.bmm(a[None], b[None]), lambda: torch.matmul(a, b)):
            with pytest.raises(S.LibraryGemmOffPrimaryStream, match="test region"):
                call()
    assert torch.mm(a, b).shape == (4, 3)                                # outside the region nothing is intercepted


def test_guard_ignores_tensors_of_other_device_types_and_integer_operands():
    a, b = torch.randn(4, 8), torch.randn(8, 3)
    with S.library_gemm_guard("assert", device_types=("cuda",)):
        assert torch.mm(a, b).shape == (4, 3)                            # host GEMM: not a device stream matter
    with S.library_gemm_guard("assert", device_types=("cpu",)):
        assert torch.mm(torch.ones(2, 2, dtype=torch.int64), torch.ones(2, 2, dtype=torch.int64)).sum() == 8


def test_guard_is_thread_local():
    import threading
    a, b = torch.randn(4, 8), torch.randn(8, 3)
    seen = {}

    def other():
        seen["y"] = torch.mm(a, b)                                       # a different host thread is not inside the region

    with S.library_gemm_guard("assert", device_types=("cpu",)):
        t = threading.Thread(target=other)
        t.start()
        t.join()
    assert seen["y"].shape == (4, 3)


def test_nt_image_of_a_right_operand():
    b = torch.randn(8, 3)
    assert S._as_nt(b).is_contiguous() and torch.equal(S._as_nt(b), b.t())
    bt = torch.randn(3, 8)
    assert S._as_nt(bt.t()).data_ptr() == bt.data_ptr()                  # a transposed view needs no copy
