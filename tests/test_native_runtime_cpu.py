"""Device-runtime C ABI (native/csrc/abi_runtime.hip -> libnd4jrocm_dev.so), host side: the sort family on host buffers against
torch.sort for every data type, the error channel, and the device queries on a box without a GPU. Reference:
libnd4j/include/legacy/NativeOps.h:1411-1441 (sort family), :150-156 (error contract); tests_cpu/layers_tests/SortCpuTests.cpp."""
import pytest
import torch

from deeplearning4j_amd.nd4j import native_runtime as R

DTYPES = (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int8, torch.uint8, torch.int16, torch.int32, torch.int64)


def _ref_sort(x, descending=False):
    v = torch.sort(x.float() if x.dtype in (torch.float16, torch.bfloat16) else x, descending=descending, stable=True).values
    return v.to(x.dtype)


@pytest.mark.parametrize("dt", DTYPES)
def test_sort_host_buffers_every_type_both_directions(dt):
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(1537, generator=g) * 40).to(dt)
    assert torch.equal(R.sort_tensor(x.clone()), _ref_sort(x))
    assert torch.equal(R.sort_tensor(x.clone(), True), _ref_sort(x, True))


def test_sort_by_key_and_by_value_move_both_arrays():
    g = torch.Generator().manual_seed(4)
    k = torch.randperm(200, generator=g).int()
    v = torch.randn(200, generator=g).double()
    k2, v2 = R.sort_by_key(k.clone(), v.clone())
    order = torch.argsort(k)
    assert torch.equal(k2, k[order]) and torch.equal(v2, v[order])
    k3, v3 = R.sort_by_key(k.clone(), v.clone(), descending=True, by_value=True)
    order = torch.argsort(v, descending=True)
    assert torch.equal(k3, k[order]) and torch.equal(v3, v[order])


def test_sort_along_every_dimension_of_dense_and_strided_arrays():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(5, 6, 7, generator=g)
    for d in range(3):
        for desc in (False, True):
            assert torch.equal(R.sort_along(x.clone(), d, desc), torch.sort(x, dim=d, descending=desc).values)
        v, i = R.sort_with_indices(x, d)
        wv, wi = torch.sort(x, dim=d, stable=True)
        assert torch.equal(v, wv) and torch.equal(i, wi)
    base = torch.randn(8, 10, generator=g)
    view = base.t()                                     # strided: the kernel walks the real strides, the base buffer is what changes
    want = torch.sort(view.clone(), dim=1).values
    R.sort_along(view, 1)
    assert torch.equal(view, want) and torch.equal(base, want.t())


def test_sort_coo_indices_orders_entries_row_major():
    idx = torch.tensor([[2, 1], [0, 3], [2, 0], [1, 1], [0, 0]], dtype=torch.int64)
    val = torch.tensor([5.0, 2.0, 4.0, 3.0, 1.0])
    R.sort_coo(idx, val, (3, 4))
    assert idx.tolist() == [[0, 0], [0, 3], [1, 1], [2, 0], [2, 1]] and val.tolist() == [1.0, 2.0, 3.0, 4.0, 5.0]


def test_error_channel_and_rejections():
    h = R.lib()
    h.dl4j_rt_clear_error()
    x = torch.randn(4, 4)
    dims = (R.C.c_int64 * 2)(0, 1)
    si = R.shape_info(x)
    h.sortTad(None, x.data_ptr(), si, None, si, dims, 2, None, None, False)          # two dimensions: not a sort along ONE dimension
    with pytest.raises(R.NativeRuntimeError, match="exactly one dimension"):
        R.check("sortTad")
    assert h.dl4j_rt_last_error_code() == 0                                           # check() cleared the slot
    with pytest.raises(AssertionError):
        R.sort_tensor(x.t())                                                          # whole-array sort needs a dense buffer
    keys, vals = torch.randn(5), torch.randn(6)
    with pytest.raises(AssertionError):
        R.sort_by_key(keys, vals)


@pytest.mark.skipif(torch.cuda.is_available(), reason="host-only expectations")
def test_device_queries_without_a_gpu():
    h = R.lib()
    assert h.getAvailableDevices() == 0 and not h.isP2PAvailable() and not h.checkP2P()
    assert h.createContext() is None
    assert not h.mallocDevice(1024, 0, 0)
    with pytest.raises(R.NativeRuntimeError, match="mallocDevice"):
        R.check("mallocDevice")
    assert h.dl4j_lc_scratch_bytes(1) == 4 << 20


def test_nd4j_sort_entry_points_on_host_arrays():
    from deeplearning4j_amd.nd4j.factory import Nd4j
    a = Nd4j.create([[3.0, 1.0, 2.0], [0.5, 4.0, -1.0]])
    assert Nd4j.sort(a, 1, True).tensor().tolist() == [[1.0, 2.0, 3.0], [-1.0, 0.5, 4.0]]
    i, v = Nd4j.sortWithIndices(a, 1, False)
    assert v.tensor().tolist() == [[3.0, 2.0, 1.0], [4.0, 0.5, -1.0]] and i.tensor().tolist() == [[0, 2, 1], [1, 0, 2]]
    from deeplearning4j_amd.nd4j.runtime_handles import NativeOpsHolder
    ops = NativeOpsHolder.getInstance().getDeviceNativeOps()
    assert ops.getAvailableDevices() == (torch.cuda.device_count() if torch.cuda.is_available() else 0)
