"""Device-runtime C ABI on the GPU: memory / streams / events / launch context through libnd4jrocm_dev.so only (raw pointers, no torch
allocation involved), and the rocPRIM sort family on device buffers against torch.sort. Reference: NativeOps.h:540-565, :597-767,
:1411-1441, :1665-1672; tests_cpu/layers_tests/SortCudaTests.cu, LaunchContextCudaTests.cu, NativeOpsTests.cpp."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from deeplearning4j_amd.nd4j import native_runtime as R  # noqa: E402

DTYPES = (torch.float32, torch.float64, torch.float16, torch.bfloat16, torch.int8, torch.uint8, torch.int16, torch.int32, torch.int64)


def dev():
    return torch.device("cuda", 0)


def test_memory_streams_events_round_trip_without_torch_buffers():
    h = R.lib()
    ops = R.DeviceNativeOps()
    assert h.getAvailableDevices() >= 1 and h.getDevice() >= 0
    p = ops.device_properties(0)
    assert p["computeUnits"] == 256 and p["wavefrontSize"] == 64 and "gfx950" in p["name"], p
    total, free = h.getDeviceTotalMemory(0), h.getDeviceFreeMemory(0)
    assert total > 200 * (1 << 30) and 0 < free <= total
    n = 1 << 20
    src = (C.c_float * n)(*([1.5] * n))
    hp, dp, dq = ops.malloc_host(4 * n), ops.malloc_device(4 * n, 0), ops.malloc_device(4 * n, 0)
    st, ev = h.createStream(), h.createEvent()
    assert st and ev
    try:
        C.memmove(hp, src, 4 * n)
        assert h.memcpyAsync(dp, hp, 4 * n, 1, st) == 1                  # host -> device on the created stream
        assert h.memcpyAsync(dq, dp, 4 * n, 3, st) == 1                  # device -> device
        assert h.memsetAsync(dp, 0, 4 * n, 0, st) == 1
        assert h.registerEvent(ev, st) == 1 and h.eventSynchronize(ev) == 1 and h.dl4j_event_query(ev) == 1
        C.memset(hp, 0xFF, 4 * n)
        assert h.memcpySync(hp, dq, 4 * n, 2, None) == 1                 # device -> host
        back = (C.c_float * n).from_address(hp)
        assert back[0] == 1.5 and back[n - 1] == 1.5
        assert h.memcpySync(hp, dp, 4 * n, 2, None) == 1
        assert back[0] == 0.0 and back[n // 2] == 0.0
        assert h.streamSynchronize(st) == 1
        assert h.getDeviceFreeMemory(0) < free                           # the two device blocks are really held
    finally:
        assert h.destroyEvent(ev) == 1 and h.dl4j_destroy_stream(st) == 1
        assert h.freeDevice(dp, 0) == 1 and h.freeDevice(dq, 0) == 1 and h.freeHost(hp) == 1
    R.check("memory round trip")


def test_launch_context_hands_out_streams_and_scratch():
    from deeplearning4j_amd.nd4j.runtime_handles import LaunchContext
    h = R.lib()
    lc = LaunchContext.defaultContext()
    a, b = lc.nativePointers(), lc.nativePointers()
    assert a == b                                                         # one context per host thread and device
    assert a["executionStream"] and a["copyStream"] and a["executionStream"] != a["copyStream"]
    assert a["scalarPointer"] and a["reductionPointer"] and a["allocationPointer"]
    # the scratch really is device memory of the advertised size: fill it on the execution stream, read the last byte back
    n = h.dl4j_lc_scratch_bytes(1)
    assert h.memsetAsync(a["reductionPointer"], 7, n, 0, a["executionStream"]) == 1 and h.streamSynchronize(a["executionStream"]) == 1
    out = (C.c_uint8 * 1)()
    assert h.memcpySync(out, a["reductionPointer"] + n - 1, 1, 2, None) == 1 and out[0] == 7


@pytest.mark.parametrize("dt", DTYPES)
def test_device_sort_every_type_both_directions(dt):
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(100003, generator=g) * 60).to(dt).to(dev())
    ref = x.float() if dt in (torch.float16, torch.bfloat16) else x
    for desc in (False, True):
        want = torch.sort(ref, descending=desc).values.to(dt)
        assert torch.equal(R.sort_tensor(x.clone(), desc), want), (dt, desc)


def test_device_sort_special_values_and_tiny_inputs():
    x = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), 3.0, -3.0, 1e-30, -1e-30], device=dev())
    got = R.sort_tensor(x.clone())
    assert torch.equal(got.cpu(), torch.tensor([-float("inf"), -3.0, -1e-30, -0.0, 0.0, 1e-30, 3.0, float("inf")]))
    assert torch.signbit(got[3]) and not torch.signbit(got[4])           # the bit-pattern order puts -0 before +0
    assert R.sort_tensor(torch.tensor([2.0], device=dev())).tolist() == [2.0]
    assert R.sort_tensor(torch.empty(0, device=dev())).numel() == 0


def test_device_sort_by_key_by_value_and_along_dimensions():
    g = torch.Generator().manual_seed(12)
    k = torch.randperm(50000, generator=g).int().to(dev())
    v = torch.randn(50000, generator=g).to(dev())
    k2, v2 = R.sort_by_key(k.clone(), v.clone())
    order = torch.argsort(k)
    assert torch.equal(k2, k[order]) and torch.equal(v2, v[order])
    k3, v3 = R.sort_by_key(k.clone(), v.clone(), descending=True, by_value=True)
    order = torch.argsort(v, descending=True)
    assert torch.equal(k3, k[order]) and torch.equal(v3, v[order])
    x = torch.randn(37, 41, 29, generator=g).to(dev())
    for d in range(3):
        for desc in (False, True):
            assert torch.equal(R.sort_along(x.clone(), d, desc), torch.sort(x, dim=d, descending=desc).values), (d, desc)
        vv, ii = R.sort_with_indices(x, d)
        assert torch.equal(vv, torch.sort(x, dim=d).values) and torch.equal(torch.gather(x, d, ii), vv)
    big = torch.randn(4, 200000, generator=g).to(dev()).bfloat16()         # long rows: the segmented sort's large-segment path
    assert torch.equal(R.sort_along(big.clone(), 1), torch.sort(big.float(), dim=1).values.bfloat16())
    base = torch.randn(64, 96, generator=g).to(dev())
    view = base.t()
    want = torch.sort(view.clone(), dim=1, descending=True).values
    R.sort_along(view, 1, True)
    assert torch.equal(view, want)


def test_device_sort_coo_and_nd4j_entry_points():
    from deeplearning4j_amd.nd4j.factory import Nd4j
    g = torch.Generator().manual_seed(13)
    shape = (50, 60, 70)
    flat = torch.randperm(50 * 60 * 70, generator=g)[:5000]
    idx = torch.stack([flat // 4200, (flat // 70) % 60, flat % 70], 1).to(dev())
    val = torch.randn(5000, generator=g).to(dev())
    i2, v2 = idx.clone(), val.clone()
    R.sort_coo(i2, v2, shape)
    order = torch.argsort(flat.to(dev()))
    assert torch.equal(i2, idx[order]) and torch.equal(v2, val[order])
    a = torch.randn(128, 333, generator=g).to(dev())
    assert torch.equal(Nd4j.sort(a, 1, True).tensor(), torch.sort(a, dim=1).values)
    i, v = Nd4j.sortWithIndices(a, 0, False)
    assert torch.equal(v.tensor(), torch.sort(a, dim=0, descending=True).values) and torch.equal(torch.gather(a, 0, i.tensor()), v.tensor())
