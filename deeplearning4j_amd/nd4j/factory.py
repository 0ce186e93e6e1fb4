"""``Nd4j`` static factory (reference: nd4j-api linalg/factory/Nd4j.java, ~7k LoC).

Backend boot: the reference resolves ~20 backend roles reflectively from a properties file
(Nd4j.initWithBackend :5210-5345). Here there is exactly one backend — MI355X through
PyTorch-ROCm + the in-tree HIP library — so the roles are plain module-level singletons:
executioner (nd4j/ops/legacy.py + ops/custom.py), workspace manager (workspace.py), memory manager,
affinity manager (one process per GPU: the affinity is LOCAL_RANK), RNG (rng.py), compressor
(compression.py).
"""
from __future__ import annotations

import ctypes
import io
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from ..native import lib as L
from . import rng as _rng
from . import serde
from .ndarray import DataType, INDArray, NDArrayIndex, _unwrap
from .ops import legacy as LG


class _State:
    dtype = torch.float32
    device = None
    order = "c"


def _dev(device=None):
    if device is not None:
        return torch.device(device)
    if _State.device is None:
        _State.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    return _State.device


def _shape(args):
    if len(args) == 1 and isinstance(args[0], (list, tuple, torch.Size)):
        return [int(s) for s in args[0]]
    return [int(s) for s in args]


class AffinityManager:
    """One process per GPU: the device of the current process is fixed at launch (LOCAL_RANK).
    (reference: nd4j-cuda CudaAffinityManager.java — round-robins JVM threads across devices)."""

    def getDeviceForCurrentThread(self):
        return torch.cuda.current_device() if torch.cuda.is_available() else 0

    def getNumberOfDevices(self):
        return torch.cuda.device_count() if torch.cuda.is_available() else 1

    def unsafeSetDevice(self, d):
        if torch.cuda.is_available():
            torch.cuda.set_device(int(d))
            _State.device = torch.device("cuda", int(d))

    attachThreadToDevice = lambda self, thread, d: self.unsafeSetDevice(d)

    def replicateToDevice(self, d, arr: INDArray):
        return INDArray(arr.tensor().to(torch.device("cuda", int(d))) if torch.cuda.is_available() else arr.tensor().clone(),
                        arr.ordering())

    def ensureLocation(self, arr, location="DEVICE"):
        return arr

    def isCrossDeviceAccessSupported(self):
        if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
            return False
        return bool(torch.cuda.can_device_access_peer(0, 1))

    def allowCrossDeviceAccess(self, b):
        pass

    def getDeviceForArray(self, arr):
        d = arr.tensor().device
        return d.index if d.type == "cuda" else -1


class MemoryManager:
    """(reference: CudaMemoryManager.java / BasicMemoryManager) — allocation is torch's caching allocator."""

    def __init__(self):
        self._periodicGc = False

    def allocate(self, nbytes, kind="DEVICE", initialize=True):
        dev = _dev() if kind == "DEVICE" else torch.device("cpu")
        t = torch.zeros(nbytes, dtype=torch.uint8, device=dev) if initialize else torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if kind == "HOST" and torch.cuda.is_available():
            t = t.pin_memory()
        return t

    def collect(self, *arrays):
        for a in arrays:
            a.close()

    def purgeCaches(self):
        if torch.cuda.is_available():
            torch.cuda.empty_cache()

    invokeGc = purgeCaches

    def memcpy(self, dst, src):
        _unwrap(dst).copy_(_unwrap(src))

    def togglePeriodicGc(self, b):
        self._periodicGc = b

    def getCurrentWorkspace(self):
        from .workspace import getWorkspaceManager
        return getWorkspaceManager().getCurrentWorkspace()

    def allocatedMemory(self, device=0):
        return torch.cuda.memory_allocated(device) if torch.cuda.is_available() else 0

    def getBandwidthUse(self):
        return {}


_AFF, _MEM = AffinityManager(), MemoryManager()


class Nd4j:
    EPS_THRESHOLD = 1e-5

    # ---- configuration
    @staticmethod
    def setDefaultDataTypes(dtype, floatDtype=None):
        _State.dtype = DataType.to_torch(floatDtype or dtype)

    @staticmethod
    def dataType():
        return DataType.from_torch(_State.dtype)

    defaultFloatingPointType = dataType

    @staticmethod
    def setDevice(device):
        _State.device = torch.device(device)

    @staticmethod
    def order():
        return _State.order

    # ---- role singletons
    @staticmethod
    def getExecutioner():
        from .ops.custom import OpExecutioner
        return OpExecutioner.getInstance()

    @staticmethod
    def getAffinityManager():
        return _AFF

    @staticmethod
    def getMemoryManager():
        return _MEM

    @staticmethod
    def getWorkspaceManager():
        from .workspace import getWorkspaceManager
        return getWorkspaceManager()

    @staticmethod
    def getRandom():
        return _rng.getRandom()

    @staticmethod
    def getRandomFactory():
        return _rng

    @staticmethod
    def getCompressor():
        from .compression import BasicNDArrayCompressor
        return BasicNDArrayCompressor.getInstance()

    @staticmethod
    def getEnvironment():
        from .environment import Environment
        return Environment.getInstance()

    @staticmethod
    def getBackend():
        return "gfx950-hip" if torch.cuda.is_available() else "cpu-torch"

    @staticmethod
    def exec(op, *a, **k):
        return Nd4j.getExecutioner().exec(op, *a, **k)

    # ---- creation
    @staticmethod
    def create(*args, dtype=None, order=None, device=None):
        """create(shape...), create(data), create(data, shape), create(dtype, shape...)"""
        order = order or _State.order
        if args and isinstance(args[0], str):
            dtype, args = args[0], args[1:]
        td = DataType.to_torch(dtype) if dtype is not None else None
        first = args[0] if args else []
        two_lists = len(args) == 2 and isinstance(args[1], (list, tuple)) and isinstance(first, (list, tuple))
        is_data = isinstance(first, (np.ndarray, torch.Tensor, INDArray)) or two_lists or (
            isinstance(first, (list, tuple)) and len(first) > 0 and any(isinstance(v, (list, tuple, float)) for v in first))
        if is_data:
            data = _unwrap(first)
            from_py = not isinstance(data, (torch.Tensor, np.ndarray))
            t = data if isinstance(data, torch.Tensor) else torch.as_tensor(np.asarray(data))
            if td is None:
                td = _State.dtype if (t.is_floating_point() and (from_py or t.dtype == torch.float64)) else t.dtype
            t = t.to(device=_dev(device), dtype=td, copy=True)
            arr = INDArray(t, "c")
            if len(args) > 1:
                arr = arr.ravel().reshape(order, _shape(args[1:]))
            return arr
        shape = _shape(args)
        t = torch.zeros(shape, dtype=td or _State.dtype, device=_dev(device))
        return INDArray(t, order)

    @staticmethod
    def createUninitialized(*shape, dtype=None, order=None, device=None):
        return INDArray(torch.empty(_shape(shape), dtype=DataType.to_torch(dtype) if dtype else _State.dtype, device=_dev(device)),
                        order or _State.order)

    createUninitializedDetached = createUninitialized

    @staticmethod
    def createFromArray(data, dtype=None, device=None):
        t = torch.as_tensor(np.asarray(data))
        if dtype is not None:
            t = t.to(DataType.to_torch(dtype))
        elif t.dtype == torch.float64:
            t = t.to(_State.dtype)
        return INDArray(t.to(_dev(device)), "c")

    @staticmethod
    def fromTensor(t: torch.Tensor, order="c"):
        return INDArray(t, order)

    @staticmethod
    def zeros(*shape, dtype=None, order=None, device=None):
        return INDArray(torch.zeros(_shape(shape), dtype=DataType.to_torch(dtype) if dtype else _State.dtype, device=_dev(device)),
                        order or _State.order)

    @staticmethod
    def ones(*shape, dtype=None, order=None, device=None):
        return INDArray(torch.ones(_shape(shape), dtype=DataType.to_torch(dtype) if dtype else _State.dtype, device=_dev(device)),
                        order or _State.order)

    @staticmethod
    def zerosLike(a):
        return a.like()

    @staticmethod
    def onesLike(a):
        return INDArray(torch.ones_like(a.tensor()), a.ordering())

    @staticmethod
    def valueArrayOf(shape, value, dtype=None, device=None):
        if isinstance(shape, int):
            shape = [shape]
        return INDArray(torch.full(_shape([shape]), value, dtype=DataType.to_torch(dtype) if dtype else _State.dtype,
                                   device=_dev(device)), _State.order)

    @staticmethod
    def scalar(value, dtype=None, device=None):
        if dtype is None:
            dtype = _State.dtype if isinstance(value, float) else (torch.bool if isinstance(value, bool) else torch.int64 if isinstance(value, int) else _State.dtype)
        return INDArray(torch.tensor(value, dtype=DataType.to_torch(dtype), device=_dev(device)), "c")

    @staticmethod
    def empty(dtype=None, device=None):
        return INDArray(torch.empty(0, dtype=DataType.to_torch(dtype) if dtype else _State.dtype, device=_dev(device)), "c")

    @staticmethod
    def eye(n, dtype=None, device=None):
        return INDArray(torch.eye(int(n), dtype=DataType.to_torch(dtype) if dtype else _State.dtype, device=_dev(device)), "c")

    @staticmethod
    def linspace(lower, upper, num, dtype=None, device=None):
        return INDArray(torch.linspace(lower, upper, int(num), dtype=DataType.to_torch(dtype) if dtype else _State.dtype,
                                       device=_dev(device)), "c")

    @staticmethod
    def arange(begin, end=None, step=1, dtype=None, device=None):
        if end is None:
            begin, end = 0, begin
        return INDArray(torch.arange(begin, end, step, dtype=DataType.to_torch(dtype) if dtype else _State.dtype,
                                     device=_dev(device)), "c")

    # ---- random
    @staticmethod
    def rand(*shape, seed=None, dtype=None, device=None, rng=None):
        r = rng or (_rng.Random(seed) if seed is not None else _rng.getRandom())
        return INDArray(r.nextFloatArray(_shape(shape), _dev(device), DataType.to_torch(dtype) if dtype else _State.dtype), _State.order)

    @staticmethod
    def randn(*shape, seed=None, dtype=None, device=None, rng=None):
        r = rng or (_rng.Random(seed) if seed is not None else _rng.getRandom())
        return INDArray(r.nextGaussianArray(_shape(shape), _dev(device), DataType.to_torch(dtype) if dtype else _State.dtype), _State.order)

    @staticmethod
    def randomBernoulli(p, *shape, device=None):
        z = torch.empty(_shape(shape), dtype=_State.dtype, device=_dev(device))
        return INDArray(_rng.exec_random("BernoulliDistribution", None, z, None, None, [p]), _State.order)

    @staticmethod
    def randomBinomial(trials, p, *shape, device=None):
        z = torch.empty(_shape(shape), dtype=_State.dtype, device=_dev(device))
        return INDArray(_rng.exec_random("BinomialDistribution", None, z, None, None, [trials, p]), _State.order)

    @staticmethod
    def randomExponential(lam, *shape, device=None):
        z = torch.empty(_shape(shape), dtype=_State.dtype, device=_dev(device))
        return INDArray(_rng.exec_random("ExponentialDistribution", None, z, None, None, [lam]), _State.order)

    @staticmethod
    def shuffle(arr: INDArray, *dims, seed=None):
        t = arr.tensor()
        g = torch.Generator().manual_seed(seed if seed is not None else _rng.getRandom().nextInt())
        d = dims[0] if dims else 0
        # DL4J shuffles the TADs along `dims`: permute the complementary axis (rows for dims=1)
        axis = 0 if (not dims or d != 0) else 1
        perm = torch.randperm(t.shape[axis], generator=g).to(t.device)
        t.copy_(t.index_select(axis, perm))
        return arr

    # ---- combination
    @staticmethod
    def concat(dim, *arrs):
        arrs = arrs[0] if len(arrs) == 1 and isinstance(arrs[0], (list, tuple)) else arrs
        return INDArray(torch.cat([_unwrap(a) for a in arrs], dim=dim), "c")

    @staticmethod
    def hstack(*arrs):
        arrs = arrs[0] if len(arrs) == 1 and isinstance(arrs[0], (list, tuple)) else arrs
        ts = [_unwrap(a) for a in arrs]
        return INDArray(torch.cat(ts, dim=0 if ts[0].dim() == 1 else 1), "c")

    @staticmethod
    def vstack(*arrs):
        arrs = arrs[0] if len(arrs) == 1 and isinstance(arrs[0], (list, tuple)) else arrs
        ts = [_unwrap(a) for a in arrs]
        ts = [t.unsqueeze(0) if t.dim() == 1 else t for t in ts]
        return INDArray(torch.cat(ts, dim=0), "c")

    @staticmethod
    def stack(dim, *arrs):
        arrs = arrs[0] if len(arrs) == 1 and isinstance(arrs[0], (list, tuple)) else arrs
        return INDArray(torch.stack([_unwrap(a) for a in arrs], dim=dim), "c")

    @staticmethod
    def pile(*arrs):
        return Nd4j.stack(0, *arrs)

    @staticmethod
    def tear(arr, *dims):
        n = arr.tensorsAlongDimension(*dims)
        return [arr.tensorAlongDimension(i, *dims).dup() for i in range(n)]

    @staticmethod
    def toFlattened(*arrs, order="c"):
        arrs = arrs[0] if len(arrs) == 1 and isinstance(arrs[0], (list, tuple)) else arrs
        return INDArray(torch.cat([(a if isinstance(a, INDArray) else INDArray(a)).ravel(order).tensor() for a in arrs]), "c")

    @staticmethod
    def pullRows(source: INDArray, sourceDimension, indexes, order="c"):
        """Rows (sourceDimension=1) or columns (0) gathered by index (NativeOps pullRows :826)."""
        t = source.tensor()
        idx = torch.as_tensor(list(indexes), dtype=torch.int64, device=t.device)
        if int(idx.min()) < 0 or int(idx.max()) >= t.shape[0 if sourceDimension == 1 else 1]:
            raise IndexError("pullRows index out of range")
        if sourceDimension != 1:
            return INDArray(t.index_select(1, idx), order)
        if t.is_cuda and t.is_contiguous() and t.dim() == 2:
            z = torch.empty((idx.numel(), t.shape[1]), dtype=t.dtype, device=t.device)
            L.check(L.hip().dl4j_pull_rows(L.ptr(t), L.ptr(idx), L.ptr(z), L.c_int64(idx.numel()), L.c_int64(t.shape[1]),
                                           L.c_int64(t.shape[0]), L.c_int(t.element_size()), L.stream()), "pull_rows")
            return INDArray(z, order)
        return INDArray(t.index_select(0, idx), order)

    @staticmethod
    def diag(a):
        return INDArray(torch.diag(_unwrap(a).reshape(-1) if _unwrap(a).dim() <= 2 and 1 in list(_unwrap(a).shape) + [1] and _unwrap(a).dim() == 1
                                   else _unwrap(a)), "c")

    @staticmethod
    def sort(arr, dim=-1, ascending=True):
        """Sorted copy along ``dim`` (Nd4j.sort -> NativeOps::sortTad; on the GPU the rocPRIM segmented radix sort of
        native/csrc/abi_runtime.hip, on the host torch's sort)."""
        t = _unwrap(arr)
        if t.is_cuda and t.dim() >= 1:
            from . import native_runtime as NR
            if NR.supported(t):
                return INDArray(NR.sort_along(t.clone(memory_format=torch.contiguous_format), dim, not ascending), "c")
        return INDArray(torch.sort(t, dim=dim, descending=not ascending).values, "c")

    @staticmethod
    def sortWithIndices(arr, dim=-1, ascending=True):
        """(indices, sorted values) along ``dim`` (Nd4j.sortWithIndices -> NativeOps::sortTadByKey with positions as the values)."""
        t = _unwrap(arr)
        if t.is_cuda and t.dim() >= 1:
            from . import native_runtime as NR
            if NR.supported(t):
                v, i = NR.sort_with_indices(t, dim, not ascending)
                return INDArray(i, "c"), INDArray(v, "c")
        v, i = torch.sort(t, dim=dim, descending=not ascending, stable=True)
        return INDArray(i, "c"), INDArray(v, "c")

    @staticmethod
    def sortCooIndices(indices, values, shape):
        """COO entries into row-major order of the full array, in place (NativeOps::sortCooIndices; BaseSparseNDArrayCOO)."""
        from . import native_runtime as NR
        NR.sort_coo(_unwrap(indices), _unwrap(values), shape)
        return indices, values

    @staticmethod
    def argMax(arr, *dims):
        return arr.argMax(*dims)

    @staticmethod
    def where(cond, x, y):
        return INDArray(torch.where(_unwrap(cond).bool(), _unwrap(x), _unwrap(y)), "c")

    @staticmethod
    def expandDims(a, dim):
        return INDArray(_unwrap(a).unsqueeze(dim), a.ordering())

    @staticmethod
    def squeeze(a, dim):
        return INDArray(_unwrap(a).squeeze(dim), a.ordering())

    @staticmethod
    def tile(a, *reps):
        return INDArray(_unwrap(a).repeat(*reps), "c")

    @staticmethod
    def reverse(a):
        return INDArray(torch.flip(_unwrap(a), dims=list(range(_unwrap(a).dim()))), "c")

    # ---- BLAS
    @staticmethod
    def gemm(a, b, c=None, transposeA=False, transposeB=False, alpha=1.0, beta=0.0):
        A, B = _unwrap(a), _unwrap(b)
        A = A.t() if transposeA else A
        B = B.t() if transposeB else B
        if c is None:
            return INDArray(alpha * (A @ B), "f")
        C = _unwrap(c)
        if beta == 0.0:
            torch.matmul(A, B, out=C) if C.is_contiguous() and alpha == 1.0 else C.copy_(alpha * (A @ B))
        else:
            C.copy_(torch.addmm(C, A, B, beta=beta, alpha=alpha))
        return c

    @staticmethod
    def matmul(a, b, transposeA=False, transposeB=False):
        return a.mmul(b, None, transposeA, transposeB)

    @staticmethod
    def tensorMmul(a, b, axes):
        return INDArray(torch.tensordot(_unwrap(a), _unwrap(b), dims=(list(axes[0]), list(axes[1]))), "c")

    @staticmethod
    def dot(a, b):
        return float(torch.dot(_unwrap(a).reshape(-1), _unwrap(b).reshape(-1)))

    @staticmethod
    def getBlasWrapper():
        return Nd4j

    @staticmethod
    def lapack():
        """LAPACK interface (linalg/api/blas/Lapack.java): getrf / geqrf / potrf / syev / gesvd / factor extraction."""
        from . import linalg
        return linalg.lapack()

    @staticmethod
    def syev(jobz, uplo, A, V):
        """BlasWrapper.syev: eigenvalues into V, eigenvectors overwrite A (jobz == 'V')."""
        from . import linalg
        return linalg.lapack().syev(jobz, uplo, A, V)

    # ---- multi-array DP primitives (NativeOps average / accumulate)
    @staticmethod
    def _multi_reduce(arrays, target, scale, propagate):
        ts = [_unwrap(a) for a in arrays]
        n = ts[0].numel()
        if any(t.numel() != n for t in ts):
            raise ValueError("arrays must have equal lengths")
        tgt = _unwrap(target) if target is not None else None
        gpu = all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in ts) and (
            tgt is None or (tgt.is_cuda and tgt.is_contiguous() and tgt.dtype == torch.float32 and tgt.data_ptr() % 16 == 0))
        if gpu and len({t.device for t in ts}) == 1:
            table = torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device=ts[0].device)
            L.check(L.hip().dl4j_multi_reduce(L.ptr(table), L.c_int(len(ts)), L.c_int64(n), L.ptr(tgt), L.c_float(scale),
                                              L.c_int(int(propagate)), L.stream()), "multi_reduce")
        else:
            # cross-device / host: reduce on the first array's device (host-side averaging fallback of
            # JCublasNDArrayFactory.java:712-768); cross-GPU production traffic uses RCCL instead (parallel/comm.py)
            acc = sum(t.to(ts[0].device).reshape(-1).double() for t in ts) * scale
            if tgt is not None:
                tgt.reshape(-1).copy_(acc.to(tgt.dtype))
            if propagate:
                for t in ts:
                    t.reshape(-1).copy_(acc.to(t.device, t.dtype))
        return target

    @staticmethod
    def averageAndPropagate(arrays, target=None):
        arrays = list(arrays)
        Nd4j._multi_reduce(arrays, target, 1.0 / len(arrays), True)
        return target if target is not None else arrays[0]

    @staticmethod
    def average(target, arrays):
        return Nd4j._multi_reduce(list(arrays), target, 1.0 / len(arrays), False)

    @staticmethod
    def accumulate(target, *arrays):
        arrays = arrays[0] if len(arrays) == 1 and isinstance(arrays[0], (list, tuple)) else arrays
        return Nd4j._multi_reduce(list(arrays), target, 1.0, False)

    # ---- serde
    @staticmethod
    def write(arr: INDArray, stream):
        serde.write_array(arr.tensor(), stream, arr.ordering())

    @staticmethod
    def read(stream) -> INDArray:
        return INDArray(serde.read_array(stream).to(_dev()), "c")

    @staticmethod
    def toByteArray(arr):
        return serde.to_bytes(arr.tensor(), arr.ordering())

    @staticmethod
    def fromByteArray(b):
        return INDArray(serde.from_bytes(b).to(_dev()), "c")

    @staticmethod
    def saveBinary(arr, path):
        with open(path, "wb") as f:
            Nd4j.write(arr, f)

    @staticmethod
    def readBinary(path):
        with open(path, "rb") as f:
            return Nd4j.read(io.BytesIO(f.read()))

    _NPY = {torch.float32: (b"<f4", 4), torch.float64: (b"<f8", 8), torch.float16: (b"<f2", 2), torch.int32: (b"<i4", 4),
            torch.int64: (b"<i8", 8), torch.int16: (b"<i2", 2), torch.int8: (b"|i1", 1), torch.uint8: (b"|u1", 1),
            torch.bool: (b"|b1", 1)}

    @staticmethod
    def writeAsNumpy(arr: INDArray, path):
        """.npy through the native writer (rt_buffer.cpp; reference cnpy, NativeOps.h:127 saveNpy)."""
        t = arr.tensor().detach().cpu().contiguous()
        if t.dtype == torch.bfloat16:
            t = t.float()
        descr, es = Nd4j._NPY[t.dtype]
        shape = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
        L.rt_check(L.rt().nd4j_npy_save(str(path).encode(), t.data_ptr(), shape, t.dim(), descr, es, 0), "npy save")

    @staticmethod
    def createFromNpyFile(path, device=None) -> INDArray:
        rt = L.rt()
        shape = (ctypes.c_int64 * 32)()
        rank, fortran = ctypes.c_int(0), ctypes.c_int(0)
        descr = ctypes.create_string_buffer(8)
        off = rt.nd4j_npy_header(str(path).encode(), shape, ctypes.byref(rank), descr, ctypes.byref(fortran))
        if off < 0:
            L.rt_check(int(-off), "npy header")
        d = descr.value.decode()
        rev = {v[0].decode(): k for k, v in Nd4j._NPY.items()}
        if d.startswith(">"):
            raise ValueError("big-endian npy not supported")
        key = d if d in rev else ("|" + d[1:] if ("|" + d[1:]) in rev else "<" + d[1:])
        dt = rev[key]
        shp = [int(shape[i]) for i in range(rank.value)]
        t = torch.empty(shp[::-1] if fortran.value else shp, dtype=dt)
        L.rt_check(rt.nd4j_npy_read_data(str(path).encode(), off, t.data_ptr(), t.numel() * t.element_size()), "npy read")
        if fortran.value:
            t = t.permute(*reversed(range(t.dim())))
        return INDArray(t.to(_dev(device)), "f" if fortran.value else "c")

    readNpy = createFromNpyFile

    # ---- misc
    @staticmethod
    def getStrides(shape, order="c"):
        rt = L.rt()
        r = len(shape)
        out = (ctypes.c_int64 * (2 * r + 4))()
        L.rt_check(rt.nd4j_shape_buffer(r, (ctypes.c_int64 * max(r, 1))(*shape), order.encode(), 0, out), "getStrides")
        return list(out)[1 + r:1 + 2 * r]

    @staticmethod
    def sizeOfDataType(dtype=None):
        return torch.empty(0, dtype=DataType.to_torch(dtype) if dtype else _State.dtype).element_size()

    @staticmethod
    def clearNans(arr):
        arr.tensor().nan_to_num_(nan=0.0)
        return arr


# ----------------------------------------------------------------------------- remaining Nd4j.java static surface
def _t_of(a):
    return a.tensor() if hasattr(a, "tensor") else torch.as_tensor(a)


def _wrap_like(t, a=None):
    return INDArray(t, a.ordering() if isinstance(a, INDArray) else _State.order)


class _EagerNamespace:
    """Nd4j.math() / nn() / cnn() / rnn() / loss() / linalg() / bitwise() / random() / image() / base(): eager op namespaces
    (reference: org.nd4j.linalg.factory.ops.ND*). Any registered custom-op name is callable: arrays in, INDArray(s) out."""

    _GENERATED = {"bitwise": "NDBitwise", "linalg": "NDLinalg", "random": "NDRandom", "image": "NDImage", "loss": "NDLoss"}

    def __init__(self, kind):
        self.kind = kind

    def _generated(self):
        """Typed surface generated from the op-definition DSL (deeplearning4j_amd/codegen), where the namespace has one."""
        cls = self._GENERATED.get(self.kind)
        if cls is None:
            return None
        import importlib
        return getattr(importlib.import_module("deeplearning4j_amd.autodiff.generated.nd_" + self.kind), cls)()

    def __getattr__(self, opName):
        if opName.startswith("_"):
            raise AttributeError(opName)
        gen = self._generated()
        if gen is not None and hasattr(gen, opName):
            fn = getattr(gen, opName)
            wrap = lambda o: INDArray(o) if isinstance(o, torch.Tensor) else o

            def typed(*a, **k):
                r = fn(*a, **k)
                return tuple(wrap(o) for o in r) if isinstance(r, tuple) else wrap(r)
            return typed
        from .ops.custom import describe, exec_op
        import re as _re
        cand = [opName, _re.sub(r"(?<!^)(?=[A-Z])", "_", opName).lower(), opName.lower()]
        name = next((c for c in cand if describe(c) is not None), None)
        if name is None:
            raise AttributeError("Nd4j.%s(): no op named '%s' in the registry" % (self.kind, opName))

        def call(*inputs, iArgs=None, tArgs=None, bArgs=None, dArgs=None):
            outs = exec_op(name, [_t_of(x) for x in inputs], iArgs=iArgs, tArgs=tArgs, bArgs=bArgs, dArgs=dArgs)
            outs = [INDArray(o) if isinstance(o, torch.Tensor) else o for o in outs]
            return outs[0] if len(outs) == 1 else outs
        return call


def _install_nd4j_extras():
    S = staticmethod
    N = Nd4j
    for ns in ("math", "nn", "cnn", "rnn", "loss", "linalg", "bitwise", "random", "image", "base"):
        setattr(N, ns, S((lambda k: (lambda: _EagerNamespace(k)))(ns)))
    # reductions along dimensions (Nd4j.sum(arr, dims...) etc.)
    for nm in ("sum", "mean", "max", "min", "prod", "norm1", "norm2", "normmax", "std", "var"):
        setattr(N, nm, S((lambda m: (lambda arr, *dims: getattr(arr, m)(*dims)))(nm)))
    N.argMin = S(lambda arr, *dims: arr.argMin(*dims) if hasattr(arr, "argMin") else INDArray(torch.argmin(_t_of(arr), dim=dims[0] if dims else None)))
    N.cumsum = S(lambda arr, dim=-1: INDArray(torch.cumsum(_t_of(arr), dim)))
    N.shape = S(lambda arr: list(_t_of(arr).shape))
    N.copy = S(lambda a, b: (_t_of(b).copy_(_t_of(a)), b)[1])
    N.append = S(lambda arr, padAmount, val, axis=-1: _wrap_like(torch.cat(
        [_t_of(arr), torch.full([padAmount if d == axis % _t_of(arr).dim() else s for d, s in enumerate(_t_of(arr).shape)], float(val),
                                dtype=_t_of(arr).dtype, device=_t_of(arr).device)], axis), arr))
    N.prepend = S(lambda arr, padAmount, val, axis=-1: _wrap_like(torch.cat(
        [torch.full([padAmount if d == axis % _t_of(arr).dim() else s for d, s in enumerate(_t_of(arr).shape)], float(val),
                    dtype=_t_of(arr).dtype, device=_t_of(arr).device), _t_of(arr)], axis), arr))

    def appendBias(*vectors):
        """Column of the concatenated vectors with a trailing 1 (bias unit)."""
        t = torch.cat([_t_of(v).reshape(-1) for v in vectors] + [torch.ones(1, dtype=_t_of(vectors[0]).dtype, device=_t_of(vectors[0]).device)])
        return INDArray(t.reshape(-1, 1))
    N.appendBias = S(appendBias)

    def pad(arr, padWidth, mode="CONSTANT", value=0.0):
        t = _t_of(arr)
        pw = [list(p) if isinstance(p, (list, tuple)) else [int(p), int(p)] for p in (padWidth if isinstance(padWidth[0], (list, tuple)) else [padWidth] * t.dim())]
        flat = [v for p in reversed(pw) for v in p]
        m = {"CONSTANT": "constant", "REFLECT": "reflect", "SYMMETRIC": "symmetric", "EDGE": "replicate"}[str(mode).upper()]
        if m == "symmetric":
            from .ops.custom import exec_op
            return INDArray(exec_op("mirror_pad", [t, torch.tensor(pw)], iArgs=[1])[0])
        if m == "constant":
            return _wrap_like(torch.nn.functional.pad(t, flat, mode=m, value=float(value)), arr)
        t2 = t.reshape((1,) * max(0, len(flat) // 2 + 1 - t.dim()) + tuple(t.shape)) if t.dim() < len(flat) // 2 + 1 else t
        return _wrap_like(torch.nn.functional.pad(t2, flat, mode=m).reshape([s + a + b for s, (a, b) in zip(t.shape, pw)]), arr)
    N.pad = S(pad)
    N.repeat = S(lambda arr, n: _wrap_like(torch.cat([_t_of(arr)] * int(n), 0), arr))
    N.rollAxis = S(lambda arr, axis, start=0: _wrap_like(torch.movedim(_t_of(arr), axis, start if start <= axis else start - 1), arr))
    N.rot90 = S(lambda arr: _t_of(arr).copy_(torch.rot90(_t_of(arr), 1, (0, 1)).reshape(_t_of(arr).shape)) if _t_of(arr).shape[0] == _t_of(arr).shape[1]
                else _wrap_like(torch.rot90(_t_of(arr), 1, (0, 1)), arr))
    N.meshgrid = S(lambda x, y: [INDArray(g) for g in torch.meshgrid(_t_of(x).reshape(-1), _t_of(y).reshape(-1), indexing="xy")])
    N.tri = S(lambda n, m=None, k=0, dtype=None: INDArray(torch.tril(torch.ones(n, m or n, dtype=DataType.to_torch(dtype) if dtype else _State.dtype,
                                                                                   device=_State.device), k)))
    N.triu = S(lambda arr, k=0: _wrap_like(torch.triu(_t_of(arr), k), arr))
    N.stripOnes = S(lambda arr: _wrap_like(_t_of(arr).reshape([s for s in _t_of(arr).shape if s != 1] or [1]), arr))

    def sortRows(arr, colIdx, ascending=True):
        t = _t_of(arr)
        return _wrap_like(t[torch.argsort(t[:, colIdx], descending=not ascending, stable=True)], arr)

    def sortColumns(arr, rowIdx, ascending=True):
        t = _t_of(arr)
        return _wrap_like(t[:, torch.argsort(t[rowIdx], descending=not ascending, stable=True)], arr)
    N.sortRows, N.sortColumns = S(sortRows), S(sortColumns)
    N.specialConcat = S(lambda dim, *arrs: N.concat(dim, *arrs))

    def choice(source, probs, numSamples_or_target, rng=None):
        """Sample elements of ``source`` with the given probabilities (Nd4j.choice -> legacy random op Choice)."""
        src, p = _t_of(source).reshape(-1), _t_of(probs).reshape(-1).double()
        n = numSamples_or_target if isinstance(numSamples_or_target, int) else _t_of(numSamples_or_target).numel()
        g = rng.generator(src.device) if (rng is not None and hasattr(rng, "generator")) else None
        idx = torch.multinomial(p / p.sum(), n, replacement=True, generator=g)
        out = src[idx.to(src.device)]
        if isinstance(numSamples_or_target, int):
            return INDArray(out)
        _t_of(numSamples_or_target).view(-1).copy_(out)
        return numSamples_or_target
    N.choice = S(choice)

    def scatterUpdate(op, array, indices, updates, axis):
        """In-place update of the sub-arrays of ``array`` along ``axis`` picked by ``indices`` (ScatterUpdate.UpdateOp order:
        ADD, SUBTRACT, MULTIPLY, DIVIDE, RSUBTRACT, RDIVIDE, ASSIGN)."""
        t, u = _t_of(array), _t_of(updates)
        ax = [a % t.dim() for a in (axis if isinstance(axis, (list, tuple)) else [axis])]
        lead = [d for d in range(t.dim()) if d not in ax]
        tv = t.permute(*lead, *ax)
        tv = tv.reshape(-1, *tv.shape[len(lead):]) if lead else tv.unsqueeze(0)
        uv = u.reshape(len(indices), *tv.shape[1:])
        k = op if isinstance(op, int) else ["ADD", "SUBTRACT", "MULTIPLY", "DIVIDE", "RSUBTRACT", "RDIVIDE", "ASSIGN"].index(str(op).upper())
        for j, i in enumerate(int(v) for v in (indices.tolist() if hasattr(indices, "tolist") else indices)):
            cur = tv[i]
            tv[i] = [cur + uv[j], cur - uv[j], cur * uv[j], cur / uv[j], uv[j] - cur, uv[j] / cur, uv[j]][k]
        return array
    N.scatterUpdate = S(scatterUpdate)
    # buffers / construction from descriptors
    from .common import DataBuffer as _DB

    def createBuffer(*args, dtype=None):
        a0 = args[0] if args else 0
        dt = DataType.to_torch(dtype) if dtype else _State.dtype
        if isinstance(a0, int):
            return _DB(torch.zeros(a0, dtype=dt))
        if isinstance(a0, (list, tuple)) and a0 and isinstance(a0[0], int) and len(args) > 1 and isinstance(args[1], str):
            return _DB(torch.zeros(int(np.prod(a0)), dtype=DataType.to_torch(args[1])))
        return _DB(torch.as_tensor(a0, dtype=dt if dtype or not isinstance(a0, torch.Tensor) else None).reshape(-1))
    N.createBuffer = N.createBufferDetached = S(createBuffer)
    N.createTypedBuffer = N.createTypedBufferDetached = S(lambda data, dtype: _DB(torch.as_tensor(data).to(DataType.to_torch(dtype)).reshape(-1)))

    def createArrayFromShapeBuffer(data, shapeInfo):
        si = [int(v) for v in (_t_of(shapeInfo).tolist() if not isinstance(shapeInfo, (list, tuple)) else shapeInfo)]
        r = si[0]
        shape, stride = si[1:1 + r], si[1 + r:1 + 2 * r]
        order = "f" if si[-1] == 102 else "c"
        base = data.tensor() if hasattr(data, "tensor") else torch.as_tensor(data)
        return INDArray(torch.as_strided(base.reshape(-1), shape, stride), order)
    N.createArrayFromShapeBuffer = S(createArrayFromShapeBuffer)

    def createFromFlatArray(buf):
        from ..autodiff import serde as ASD
        return INDArray(ASD.flat_array_from_bytes(buf)) if hasattr(ASD, "flat_array_from_bytes") else INDArray(serde.from_bytes(buf))
    N.createFromFlatArray = S(createFromFlatArray)

    # numpy / text IO
    def toNpyByteArray(arr):
        import io as _io
        b = _io.BytesIO()
        np.save(b, _t_of(arr).detach().cpu().numpy(), allow_pickle=False)
        return b.getvalue()

    def createNpyFromByteArray(data):
        import io as _io
        return INDArray(torch.from_numpy(np.load(_io.BytesIO(bytes(data)), allow_pickle=False)).to(_State.device))
    N.toNpyByteArray, N.convertToNumpy = S(toNpyByteArray), S(toNpyByteArray)
    N.createNpyFromByteArray = S(createNpyFromByteArray)
    N.createNpyFromInputStream = S(lambda stream: createNpyFromByteArray(stream.read()))
    N.createFromNpyPointer = S(lambda pointer: createNpyFromByteArray(pointer))

    def writeTxt(arr, path):
        """The text format of Nd4j.writeTxt: a JSON header (filefrom, ordering, shape, data type) followed by the values."""
        import json
        t = _t_of(arr).detach().cpu()
        with open(str(path), "w") as f:
            f.write(json.dumps({"filefrom": "dl4j", "ordering": arr.ordering() if isinstance(arr, INDArray) else "c",
                                "shape": list(t.shape), "dataType": DataType.from_torch(t.dtype)}) + "\n")
            f.write(json.dumps(t.double().tolist() if t.is_floating_point() else t.tolist()))

    def readTxtString(text):
        import json
        head, _, body = text.partition("\n")
        h = json.loads(head)
        t = torch.tensor(json.loads(body), dtype=DataType.to_torch(h["dataType"])).reshape(h["shape"])
        return INDArray(t.to(_State.device), h.get("ordering", "c"))
    N.writeTxt, N.readTxtString = S(writeTxt), S(readTxtString)
    N.readTxt = S(lambda path: readTxtString(open(str(path)).read()))

    def readNumpy(path, sep=",", dtype=None):
        """Delimited text matrix (numpy.savetxt style), parsed by the native CSV reader."""
        rows = [[float(v) for v in line.strip().split(sep)] for line in open(str(path)) if line.strip()]
        return INDArray(torch.tensor(rows, dtype=DataType.to_torch(dtype) if dtype else _State.dtype, device=_State.device))
    N.readNumpy = S(readNumpy)
    # randomness
    N.randWithSeed = S(lambda seed, *shape: INDArray(torch.rand(*[int(s) for s in (shape[0] if len(shape) == 1 and isinstance(shape[0], (list, tuple)) else shape)],
                                                                generator=torch.Generator().manual_seed(int(seed)), dtype=_State.dtype).to(_State.device)))
    N.getDistributions = S(lambda: __import__("deeplearning4j_amd.nd4j.rng", fromlist=["Distributions"]).Distributions
                           if hasattr(__import__("deeplearning4j_amd.nd4j.rng", fromlist=["x"]), "Distributions") else N.getRandom())
    # environment switches / role singletons
    from . import environment as _env
    N.setDataType = S(lambda dt: N.setDefaultDataTypes(dt, dt))
    N.allowsSpecifyOrdering = S(lambda: True)
    N.checkShapeValues = S(lambda *shape: [int(v) for v in (shape[0] if len(shape) == 1 and isinstance(shape[0], (list, tuple)) else shape)]
                           if all(int(v) >= 0 for v in (shape[0] if len(shape) == 1 and isinstance(shape[0], (list, tuple)) else shape))
                           else (_ for _ in ()).throw(ValueError("Invalid shape: Requested INDArray shape contains dimension size values < 0")))
    N.enableFallbackMode = S(lambda flag: setattr(_State, "fallback", bool(flag)))
    N.isFallbackModeEnabled = S(lambda: bool(getattr(_State, "fallback", False)))
    N.isExperimentalMode = S(lambda: False)
    N.isPrecisionBoostAllowed = S(lambda: bool(L.rt().env_get_precision_boost()))
    N.getProfiler = S(lambda: _env.OpProfiler.getInstance())
    N.toggleTrace = S(lambda flag: L.rt().op_trace_toggle(int(bool(flag))))
    N.purgeTrace = S(lambda: L.rt().op_trace_purge())
    N.factory = N.getNDArrayFactory = S(lambda: N)
    N.setFactory = S(lambda f: None)
    N.getDataBufferFactory = S(lambda: N)
    N.getBlasLapackDelegator = S(lambda: N.getBlasWrapper() if hasattr(N, "getBlasWrapper") else N)
    N.getConstantHandler = N.getShapeInfoProvider = S(lambda: L.rt())
    N.getDeviceIdProvider = S(lambda: N.getAffinityManager())
    N.getStatsProvider = S(lambda: N.getMemoryManager())

    def getDeallocatorService():
        from .backend import DeallocatorService
        return DeallocatorService.getInstance() if hasattr(DeallocatorService, "getInstance") else DeallocatorService()
    N.getDeallocatorService = S(getDeallocatorService)
    N._convolution = None
    N.getConvolution = S(lambda: N._convolution or _EagerNamespace("cnn"))
    N.setConvolution = S(lambda c: setattr(N, "_convolution", c))

    def close(*arrs):
        for a in arrs:
            if a is not None and hasattr(a, "close"):
                a.close()
    N.close = S(close)


_install_nd4j_extras()
