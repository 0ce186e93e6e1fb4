"""Device-runtime groups of the C ABI (memory, devices, streams / events, P2P, launch context, sort family) — ctypes binding of
``libnd4jrocm_dev.so`` (native/csrc/abi_runtime.hip) under the reference's names (nd4j-api/.../nativeblas/NativeOps.java mirrors
libnd4j/include/legacy/NativeOps.h :540-565, :597-699, :710-767, :856-867, :1411-1441, :1665-1672).

The library is self-contained on the HIP runtime and rocPRIM: a foreign binding (the JavaCPP preset the reference ships) can allocate,
copy, order streams and sort without torch in the process. Inside this package the same entry points serve ``Nd4j.sort`` /
``sortWithIndices`` / ``sortCooIndices`` on GPU arrays (``sort_tensor`` & co. below run them on torch's current stream) and the
reference-named handle objects of nd4j/runtime_handles.py.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import torch

HERE = Path(__file__).resolve().parent.parent / "native"
_LIB = None

_FLAG = {torch.bfloat16: 2048, torch.float16: 4096, torch.float32: 8192, torch.float64: 16384, torch.int8: 32768, torch.int16: 65536,
         torch.int32: 131072, torch.int64: 262144, torch.bool: 524288, torch.uint8: 32768 | 8388608}
P64 = C.POINTER(C.c_int64)
VPP = C.POINTER(C.c_void_p)


class NativeRuntimeError(RuntimeError):
    pass


def lib():
    """The loaded library (RTLD_LOCAL: its exports carry the reference's unprefixed names, e.g. ``sort`` / ``setDevice``)."""
    global _LIB
    if _LIB is None:
        path = HERE / "libnd4jrocm_dev.so"
        if not path.exists():
            from ..native import build as _b
            _b.build()
        try:
            h = C.CDLL(str(path), mode=C.RTLD_LOCAL)
        except OSError as e:
            raise NativeRuntimeError("libnd4jrocm_dev.so could not be loaded (%s); build it with `python -m deeplearning4j_amd.native.build`" % e)
        vp, i32, i64, b = C.c_void_p, C.c_int, C.c_int64, C.c_bool
        sig = {
            "dl4j_rt_last_error_code": (i32, []), "dl4j_rt_last_error_message": (C.c_char_p, []), "dl4j_rt_clear_error": (None, []),
            "mallocHost": (vp, [i64, i32]), "mallocDevice": (vp, [i64, i32, i32]), "freeHost": (i32, [vp]), "freeDevice": (i32, [vp, i32]),
            "memcpySync": (i32, [vp, vp, i64, i32, vp]), "memcpyAsync": (i32, [vp, vp, i64, i32, vp]),
            "memsetSync": (i32, [vp, i32, i64, i32, vp]), "memsetAsync": (i32, [vp, i32, i64, i32, vp]),
            "getAvailableDevices": (i32, []), "setDevice": (i32, [i32]), "getDevice": (i32, []),
            "getDeviceFreeMemory": (i64, [i32]), "getDeviceFreeMemoryDefault": (i64, []), "getDeviceTotalMemory": (i64, [i32]),
            "getDeviceMajor": (i32, [i32]), "getDeviceMinor": (i32, [i32]), "getDeviceName": (C.c_char_p, [i32]),
            "dl4j_device_properties": (i32, [i32, P64]),
            "isP2PAvailable": (b, []), "enableP2P": (None, [b]), "checkP2P": (b, []),
            "createContext": (vp, []), "createStream": (vp, []), "dl4j_destroy_stream": (i32, [vp]), "createEvent": (vp, []),
            "registerEvent": (i32, [vp, vp]), "destroyEvent": (i32, [vp]), "streamSynchronize": (i32, [vp]), "eventSynchronize": (i32, [vp]),
            "dl4j_stream_wait_event": (i32, [vp, vp]), "dl4j_event_query": (i32, [vp]),
            "defaultLaunchContext": (vp, []), "lcScalarPointer": (vp, [vp]), "lcReductionPointer": (vp, [vp]), "lcAllocationPointer": (vp, [vp]),
            "lcExecutionStream": (vp, [vp]), "lcCopyStream": (vp, [vp]), "lcBlasHandle": (vp, [vp]), "lcSolverHandle": (vp, [vp]),
            "dl4j_lc_scratch_bytes": (i64, [i32]),
            "sort": (None, [VPP, vp, P64, vp, P64, b]),
            "sortByKey": (None, [VPP, vp, P64, vp, P64, vp, P64, vp, P64, b]),
            "sortByValue": (None, [VPP, vp, P64, vp, P64, vp, P64, vp, P64, b]),
            "sortTad": (None, [VPP, vp, P64, vp, P64, P64, i64, P64, P64, b]),
            "sortTadByKey": (None, [VPP, vp, P64, vp, P64, vp, P64, vp, P64, P64, C.c_longlong, b]),
            "sortTadByValue": (None, [VPP, vp, P64, vp, P64, vp, P64, vp, P64, P64, i64, b]),
            "sortCooIndices": (None, [VPP, P64, vp, i64, P64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _LIB = h
    return _LIB


def check(what="native runtime"):
    """Raise the library's recorded error (the reference's lastErrorCode / lastErrorMessage contract) and clear it."""
    h = lib()
    code = h.dl4j_rt_last_error_code()
    if code:
        msg = (h.dl4j_rt_last_error_message() or b"").decode(errors="replace")
        h.dl4j_rt_clear_error()
        raise NativeRuntimeError("%s failed (%d): %s" % (what, code, msg))


def shape_info(t: torch.Tensor):
    """shapeInfo = [rank, shape.., strides.., extras(dtype flags), ews, order] with the tensor's REAL strides."""
    r = t.dim()
    si = (C.c_int64 * (2 * r + 4))()
    si[0] = r
    for k in range(r):
        si[1 + k] = int(t.shape[k])
        si[1 + r + k] = int(t.stride(k))
    si[1 + 2 * r] = _FLAG[t.dtype]
    si[2 + 2 * r] = 1 if t.is_contiguous() else 0
    si[3 + 2 * r] = 99          # 'c'
    return si


def _extra(t: torch.Tensor):
    """extraPointers for a call on ``t``'s device: [0] = "device buffers" marker, [1] = torch's current stream."""
    if t.is_cuda:
        return (C.c_void_p * 2)(C.c_void_p(1), C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream))
    return (C.c_void_p * 2)(None, None)


def _pair(t: torch.Tensor):
    p = C.c_void_p(t.data_ptr())
    return (None, p) if t.is_cuda else (p, None)


def supported(t: torch.Tensor) -> bool:
    return isinstance(t, torch.Tensor) and t.dtype in _FLAG and t.numel() < (1 << 32)


def sort_tensor(t: torch.Tensor, descending=False):
    """In-place sort of a dense tensor's elements in memory order (NativeOps::sort)."""
    assert t.is_contiguous()
    si = shape_info(t)
    h, d = _pair(t)
    lib().sort(_extra(t), h, si, d, si, bool(descending))
    check("sort")
    return t


def sort_by_key(keys: torch.Tensor, values: torch.Tensor, descending=False, by_value=False):
    """In-place: both arrays reordered by ``keys`` (sortByKey) or by ``values`` (sortByValue)."""
    assert keys.is_contiguous() and values.is_contiguous() and keys.numel() == values.numel() and keys.device == values.device
    ks, vs = shape_info(keys), shape_info(values)
    kh, kd = _pair(keys)
    vh, vd = _pair(values)
    fn = lib().sortByValue if by_value else lib().sortByKey
    fn(_extra(keys), kh, ks, kd, ks, vh, vs, vd, vs, bool(descending))
    check("sortByValue" if by_value else "sortByKey")
    return keys, values


def sort_along(t: torch.Tensor, dim=-1, descending=False, values: Optional[torch.Tensor] = None, by_value=False):
    """In-place sort of every sub-array along ``dim`` (sortTad); with ``values`` the pair form (sortTadByKey / sortTadByValue).
    Works on strided views: the kernel walks the real strides."""
    dims = (C.c_int64 * 1)(int(dim) % max(t.dim(), 1))
    si = shape_info(t)
    h, d = _pair(t)
    if values is None:
        lib().sortTad(_extra(t), h, si, d, si, dims, 1, None, None, bool(descending))
        check("sortTad")
        return t
    assert values.shape == t.shape and values.device == t.device
    vs = shape_info(values)
    vh, vd = _pair(values)
    fn = lib().sortTadByValue if by_value else lib().sortTadByKey
    fn(_extra(t), h, si, d, si, vh, vs, vd, vs, dims, 1, bool(descending))
    check("sortTadByValue" if by_value else "sortTadByKey")
    return t, values


def sort_with_indices(t: torch.Tensor, dim=-1, descending=False):
    """(sorted copy, int64 indices) along ``dim`` through sortTadByKey with a position array as the values."""
    dim = int(dim) % max(t.dim(), 1)
    out = t.clone(memory_format=torch.contiguous_format)
    shape = [1] * t.dim()
    shape[dim] = t.shape[dim]
    idx = torch.arange(t.shape[dim], device=t.device, dtype=torch.int64).view(shape).expand(t.shape).contiguous()
    sort_along(out, dim, descending, values=idx)
    return out, idx


def sort_coo(indices: torch.Tensor, values: torch.Tensor, shape):
    """In-place: COO entries (indices [nnz, rank] int64, values [nnz]) into row-major order of the full ``shape`` (sortCooIndices)."""
    assert indices.dtype == torch.int64 and indices.is_contiguous() and values.is_contiguous() and indices.device == values.device
    r = len(shape)
    si = (C.c_int64 * (2 * r + 4))()
    si[0] = r
    st = 1
    for k in range(r - 1, -1, -1):
        si[1 + k] = int(shape[k])
        si[1 + r + k] = st
        st *= int(shape[k])
    si[1 + 2 * r], si[2 + 2 * r], si[3 + 2 * r] = _FLAG[values.dtype], 1, 99
    lib().sortCooIndices(_extra(values), C.cast(C.c_void_p(indices.data_ptr()), P64), C.c_void_p(values.data_ptr()), int(values.numel()), si)
    check("sortCooIndices")
    return indices, values


class DeviceNativeOps:
    """The reference's ``NativeOps`` device-runtime methods (same names and argument order) over the native library."""

    def __init__(self):
        self.lib = lib()

    def __getattr__(self, name):            # every declared entry point is reachable under its ABI name
        return getattr(self.lib, name)

    # convenience forms that raise on failure
    def malloc_device(self, nbytes: int, device: int = -1) -> int:
        p = self.lib.mallocDevice(int(nbytes), int(device), 0)
        if not p:
            check("mallocDevice")
        return int(p)

    def malloc_host(self, nbytes: int) -> int:
        p = self.lib.mallocHost(int(nbytes), 0)
        if not p:
            check("mallocHost")
        return int(p)

    def device_properties(self, device: int = 0) -> dict:
        out = (C.c_int64 * 6)()
        if self.lib.dl4j_device_properties(int(device), out):
            check("deviceProperties")
        keys = ("computeUnits", "wavefrontSize", "ldsPerWorkgroup", "l2Bytes", "memoryClockKHz", "memoryBusWidth")
        d = dict(zip(keys, [int(v) for v in out]))
        d.update(name=self.lib.getDeviceName(int(device)).decode(), major=self.lib.getDeviceMajor(int(device)),
                 minor=self.lib.getDeviceMinor(int(device)), totalMemory=int(self.lib.getDeviceTotalMemory(int(device))))
        return d
