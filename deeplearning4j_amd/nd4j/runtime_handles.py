"""Runtime handle objects under the reference's names: ``LaunchContext`` (libnd4j include/execution/LaunchContext.h: the
stream / device a kernel launch uses), ``TadPack`` (include/array/TadPack.h: the tensors-along-dimension decomposition
of an array: sub-array shape, strides and offsets), ``AtomicAllocator`` (nd4j-cuda jita/allocator/impl/AtomicAllocator.java:
device memory bookkeeping) and ``NativeOpsHolder`` (nativeblas/NativeOpsHolder.java: the loaded native library).

Here a launch context is the current HIP stream of torch (every in-tree kernel is launched on it, native/lib.py
``stream()``), memory belongs to torch's caching HIP allocator plus the arena workspaces, and the native libraries are
the two ctypes handles of native/lib.py; these classes expose that state through the reference's accessors.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .ndarray import INDArray


class LaunchContext:
    _default = None

    def __init__(self, device=None):
        self._device = device

    @classmethod
    def defaultContext(cls) -> "LaunchContext":
        if cls._default is None:
            cls._default = LaunchContext()
        return cls._default

    def getDeviceId(self) -> int:
        if self._device is not None:
            return int(self._device)
        return torch.cuda.current_device() if torch.cuda.is_available() else 0

    def getStream(self):
        """The HIP stream kernels are launched on (None on a host-only build)."""
        return torch.cuda.current_stream(self.getDeviceId()) if torch.cuda.is_available() else None

    getCudaStream = getStream

    def getStreamHandle(self) -> int:
        s = self.getStream()
        return 0 if s is None else int(s.cuda_stream)

    def syncStream(self):
        s = self.getStream()
        if s is not None:
            s.synchronize()

    def isInitialized(self) -> bool:
        return True

    # -- the native launch context of the C ABI (NativeOps::defaultLaunchContext + lc* accessors): execution / copy streams and
    #    the scalar / reduction / allocation scratch buffers a foreign binding launches the legacy kernels with
    def nativeHandle(self) -> int:
        from .native_runtime import lib, check
        h = lib().defaultLaunchContext()
        if not h:
            check("defaultLaunchContext")
        return int(h)

    def nativePointers(self) -> dict:
        from .native_runtime import lib
        L, h = lib(), self.nativeHandle()
        return {"executionStream": L.lcExecutionStream(h), "copyStream": L.lcCopyStream(h), "scalarPointer": L.lcScalarPointer(h),
                "reductionPointer": L.lcReductionPointer(h), "allocationPointer": L.lcAllocationPointer(h),
                "blasHandle": L.lcBlasHandle(h), "solverHandle": L.lcSolverHandle(h)}


def _tt(x):
    return x.tensor() if isinstance(x, INDArray) else torch.as_tensor(x)


class TadPack:
    """Decomposition of ``array`` into the sub-arrays spanning ``dimensions`` (remaining dimensions enumerated in c order)."""

    def __init__(self, array, dimensions: Sequence[int]):
        t = _tt(array)
        r = t.dim()
        self._t = t
        self.dimensions = sorted({int(d) % r for d in dimensions})
        self._outer = [k for k in range(r) if k not in self.dimensions]
        self._oshape = [t.shape[k] for k in self._outer]

    def numberOfTads(self) -> int:
        n = 1
        for s in self._oshape:
            n *= s
        return n

    def tadShape(self) -> List[int]:
        return [self._t.shape[d] for d in self.dimensions]

    def tadStrides(self) -> List[int]:
        return [self._t.stride(d) for d in self.dimensions]

    def tadLength(self) -> int:
        n = 1
        for s in self.tadShape():
            n *= s
        return n

    def tadOffsets(self) -> List[int]:
        """Element offset of every sub-array from the start of the array's buffer view."""
        if not self._oshape:
            return [0]
        idx = np.indices(self._oshape).reshape(len(self._oshape), -1)
        st = np.array([self._t.stride(k) for k in self._outer], dtype=np.int64).reshape(-1, 1)
        return (idx * st).sum(0).astype(np.int64).tolist()

    def tad(self, index: int) -> INDArray:
        if not 0 <= index < self.numberOfTads():
            raise IndexError("TAD index %d out of range (%d)" % (index, self.numberOfTads()))
        coords = np.unravel_index(int(index), self._oshape) if self._oshape else ()
        key = [slice(None)] * self._t.dim()
        for k, c in zip(self._outer, coords):
            key[k] = int(c)
        return INDArray(self._t[tuple(key)])


class _FlowController:
    def getSpecialStream(self):
        return LaunchContext.defaultContext().getStream()

    def commitTransfer(self):
        LaunchContext.defaultContext().syncStream()


class AtomicAllocator:
    _inst = None

    @classmethod
    def getInstance(cls) -> "AtomicAllocator":
        if cls._inst is None:
            cls._inst = AtomicAllocator()
        return cls._inst

    def getDeviceId(self) -> int:
        return LaunchContext.defaultContext().getDeviceId()

    def getFlowController(self):
        return _FlowController()

    def getPointer(self, array) -> int:
        return int(_tt(array).data_ptr())

    getHostPointer = getPointer

    def synchronizeHostData(self, array):
        """Make device writes to ``array`` visible to the host (waits for the array's stream work)."""
        if _tt(array).is_cuda:
            torch.cuda.current_stream(_tt(array).device).synchronize()

    def getMemoryStats(self, device=None) -> dict:
        if not torch.cuda.is_available():
            return {"allocated": 0, "reserved": 0, "peakAllocated": 0, "free": 0, "total": 0}
        d = self.getDeviceId() if device is None else device
        free, total = torch.cuda.mem_get_info(d)
        return {"allocated": torch.cuda.memory_allocated(d), "reserved": torch.cuda.memory_reserved(d),
                "peakAllocated": torch.cuda.max_memory_allocated(d), "free": free, "total": total}

    def getTotalAllocatedDeviceMemory(self, device=None) -> int:
        return self.getMemoryStats(device)["allocated"]

    def purgeCaches(self):
        if torch.cuda.is_available():
            torch.cuda.empty_cache()


class NativeOpsHolder:
    _inst = None

    @classmethod
    def getInstance(cls) -> "NativeOpsHolder":
        if cls._inst is None:
            cls._inst = NativeOpsHolder()
        return cls._inst

    def getDeviceNativeOps(self):
        """The device-runtime ABI (memory / devices / streams / events / launch context / sort) under the reference's method names
        (nativeblas/NativeOps.java), served by libnd4jrocm_dev.so."""
        from .native_runtime import DeviceNativeOps
        return DeviceNativeOps()

    def getKernelLibrary(self):
        """ctypes handle of the HIP kernel library (libdl4j_hip.so); raises when it was not built."""
        from ..native import lib as L
        return L.hip()

    def getRuntimeNativeOps(self):
        """ctypes handle of the host runtime library (libnd4jrocm_rt.so)."""
        from ..native import lib as L
        return L.rt()
