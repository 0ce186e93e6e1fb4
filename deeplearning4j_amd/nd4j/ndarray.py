"""INDArray + NDArrayIndex: the DL4J tensor API on top of torch storage.

Reference: nd4j-api linalg/api/ndarray/{INDArray,BaseNDArray}.java (~700 methods), linalg/indexing/
NDArrayIndex.java, JCublasNDArray.java. Semantics kept: 'c'/'f' ordering (reshape/ravel follow the
array's order), views alias their parent, ``xxxi`` methods mutate in place, row/column-vector
broadcasts, TAD access, shapeInfo descriptor.

MI355X-first design: an INDArray is a thin handle over a ``torch.Tensor`` that lives in HBM (or
host); there is no dual host/device buffer with tick counters to keep in sync (DataBuffer.h:39-54)
— host access is an explicit ``.cpu()``/``toXxx`` copy. Element-wise math goes through the legacy-op
executioner (hand-written gfx950 kernels on GPU tensors), GEMMs through hipBLASLt via torch.
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from ..native import lib as L
from . import serde
from .ops import legacy as LG

_DT = {"DOUBLE": torch.float64, "FLOAT": torch.float32, "HALF": torch.float16, "BFLOAT16": torch.bfloat16,
       "LONG": torch.int64, "INT": torch.int32, "SHORT": torch.int16, "BYTE": torch.int8, "UBYTE": torch.uint8,
       "BOOL": torch.bool, "FLOAT16": torch.float16, "INT64": torch.int64, "INT32": torch.int32, "FLOAT32": torch.float32,
       "FLOAT64": torch.float64, "INT8": torch.int8, "INT16": torch.int16, "UINT8": torch.uint8}
_DT_NAME = {torch.float64: "DOUBLE", torch.float32: "FLOAT", torch.float16: "HALF", torch.bfloat16: "BFLOAT16",
            torch.int64: "LONG", torch.int32: "INT", torch.int16: "SHORT", torch.int8: "BYTE", torch.uint8: "UBYTE",
            torch.bool: "BOOL"}


class DataType:
    DOUBLE, FLOAT, HALF, BFLOAT16, LONG, INT, SHORT, BYTE, UBYTE, BOOL = (
        "DOUBLE", "FLOAT", "HALF", "BFLOAT16", "LONG", "INT", "SHORT", "BYTE", "UBYTE", "BOOL")
    FLOAT16, INT64, INT32 = "HALF", "LONG", "INT"

    @staticmethod
    def to_torch(dt):
        if isinstance(dt, torch.dtype):
            return dt
        return _DT[str(dt).upper()]

    @staticmethod
    def from_torch(dt):
        return _DT_NAME[dt]


def _ex():
    return LG.executioner()


def _unwrap(x):
    return x._t if isinstance(x, INDArray) else x


# ----------------------------------------------------------------------------- indexing
class _Idx:
    pass


class _All(_Idx):
    pass


class _Point(_Idx):
    def __init__(self, i):
        self.i = int(i)


class _Interval(_Idx):
    def __init__(self, b, e, stride=1, inclusive=False):
        self.b, self.e, self.s = int(b), int(e) + (1 if inclusive else 0), int(stride)


class _Indices(_Idx):
    def __init__(self, idx):
        self.idx = [int(i) for i in idx]


class _NewAxis(_Idx):
    pass


class NDArrayIndex:
    @staticmethod
    def all():
        return _All()

    @staticmethod
    def point(i):
        return _Point(i)

    @staticmethod
    def interval(b, e_or_stride, e=None, inclusive=False):
        if e is None:
            return _Interval(b, e_or_stride, 1, inclusive)
        return _Interval(b, e, e_or_stride, inclusive)

    @staticmethod
    def indices(*idx):
        if len(idx) == 1 and isinstance(idx[0], (list, tuple)):
            idx = idx[0]
        return _Indices(idx)

    @staticmethod
    def newAxis():
        return _NewAxis()

    @staticmethod
    def _to_key(indexes, rank):
        key = []
        for ix in indexes:
            if isinstance(ix, _All):
                key.append(slice(None))
            elif isinstance(ix, _Point):
                key.append(ix.i)
            elif isinstance(ix, _Interval):
                key.append(slice(ix.b, ix.e, ix.s))
            elif isinstance(ix, _Indices):
                key.append(ix.idx)
            elif isinstance(ix, _NewAxis):
                key.append(None)
            elif isinstance(ix, (int, slice)):
                key.append(ix)
            else:
                raise TypeError("unsupported index " + repr(ix))
        return tuple(key)


# ----------------------------------------------------------------------------- INDArray
class INDArray:
    __slots__ = ("_t", "_order", "_closed", "_ws")
    __array_priority__ = 100

    def __init__(self, t: torch.Tensor, order: str = "c"):
        self._t = t
        self._order = order
        self._closed = False
        self._ws = None

    # -- raw access
    def tensor(self) -> torch.Tensor:
        return self._t

    def numpy(self):
        t = self._t.detach().cpu()
        return (t.float() if t.dtype == torch.bfloat16 else t).numpy()

    def data(self):
        return self._t.reshape(-1) if self._t.is_contiguous() else self._t.contiguous().reshape(-1)

    def device(self):
        return self._t.device

    def to(self, device):
        return INDArray(self._t.to(device), self._order)

    # -- shape info
    def shape(self):
        return list(self._t.shape)

    def stride(self, i=None):
        return list(self._t.stride()) if i is None else self._t.stride(i)

    def rank(self):
        return self._t.dim()

    def length(self):
        return self._t.numel()

    def size(self, i):
        return self._t.shape[i]

    def rows(self):
        return self._t.shape[0] if self._t.dim() >= 2 else 1

    def columns(self):
        return self._t.shape[-1] if self._t.dim() >= 1 else 1

    def ordering(self):
        return self._order

    def offset(self):
        return self._t.storage_offset()

    def dataType(self):
        return _DT_NAME[self._t.dtype]

    def isEmpty(self):
        return self._t.numel() == 0

    def isScalar(self):
        return self._t.numel() == 1 and self._t.dim() <= 2 and all(s == 1 for s in self._t.shape)

    def isVector(self):
        return self._t.dim() == 1 or (self._t.dim() == 2 and 1 in self._t.shape)

    def isRowVector(self):
        return self._t.dim() == 1 or (self._t.dim() == 2 and self._t.shape[0] == 1)

    def isColumnVector(self):
        return self._t.dim() == 2 and self._t.shape[1] == 1

    def isMatrix(self):
        return self._t.dim() == 2

    def isSquare(self):
        return self._t.dim() == 2 and self._t.shape[0] == self._t.shape[1]

    def isView(self):
        t = self._t
        return t._base is not None or t.storage_offset() != 0 or t.numel() * t.element_size() != t.untyped_storage().nbytes()

    def isR(self):
        return self._t.is_floating_point()

    def isZ(self):
        return not self._t.is_floating_point() and self._t.dtype != torch.bool

    def isB(self):
        return self._t.dtype == torch.bool

    def shapeInfo(self):
        """LongType[2*rank+4] descriptor built by the native runtime (rt_shape.cpp)."""
        rt = L.rt()
        r = self._t.dim()
        out = (ctypes.c_int64 * (2 * r + 4))()
        sh = (ctypes.c_int64 * max(r, 1))(*self._t.shape)
        st = (ctypes.c_int64 * max(r, 1))(*self._t.stride())
        flag = serde._FLAG[_DT_NAME[self._t.dtype]] | (8 if self._t.numel() == 0 else 0)
        L.rt_check(rt.nd4j_shape_buffer_ex(r, sh, st, self._order.encode(), flag, out), "shapeInfo")
        return list(out)

    def elementWiseStride(self):
        return self.shapeInfo()[-2]

    def shapeInfoToString(self):
        si = self.shapeInfo()
        r = si[0]
        return "Rank: %d, DataType: %s, Offset: %d, Order: %s, Shape: %s, Stride: %s" % (
            r, self.dataType(), self.offset(), self._order, si[1:1 + r], si[1 + r:1 + 2 * r])

    # -- copies / casts
    def dup(self, order=None):
        order = order or self._order
        return INDArray(self._t.clone(memory_format=torch.contiguous_format), order)

    def castTo(self, dt):
        return INDArray(self._t.to(DataType.to_torch(dt)), self._order)

    def assign(self, other):
        o = _unwrap(other)
        if isinstance(o, torch.Tensor):
            self._t.copy_(o.to(self._t.device) if o.shape == self._t.shape else o.to(self._t.device).expand(self._t.shape))
        else:
            self._t.fill_(o)
        return self

    def like(self):
        return INDArray(torch.zeros_like(self._t), self._order)

    def ulike(self):
        return INDArray(torch.empty_like(self._t), self._order)

    def detach(self):
        out = INDArray(self._t.clone(), self._order)
        return out

    def leverage(self):
        return self.detach()

    def leverageTo(self, workspaceId, enforceExistence=False):
        return self.detach()

    def migrate(self, detachOnNoWs=False):
        return self

    def isAttached(self):
        return self._ws is not None

    def close(self):
        self._closed = True
        self._t = torch.empty(0, dtype=self._t.dtype, device=self._t.device)

    def closeable(self):
        return not self._closed and not self.isView()

    def wasClosed(self):
        return self._closed

    # -- ordering-aware reshape/ravel
    def reshape(self, *args):
        order = self._order
        if args and isinstance(args[0], str):
            order, args = args[0], args[1:]
        shape = list(args[0]) if len(args) == 1 and isinstance(args[0], (list, tuple)) else list(args)
        if order == "f" and self._t.dim() > 1 or (order == "f" and len(shape) > 1):
            rev = self._t.permute(*reversed(range(self._t.dim())))
            out = rev.reshape(list(reversed(shape))).permute(*reversed(range(len(shape))))
            return INDArray(out, "f")
        return INDArray(self._t.reshape(shape), order)

    def ravel(self, order=None):
        order = order or self._order
        if order == "f" and self._t.dim() > 1:
            return INDArray(self._t.permute(*reversed(range(self._t.dim()))).reshape(-1), "c")
        return INDArray(self._t.reshape(-1), "c")

    def permute(self, *dims):
        if len(dims) == 1 and isinstance(dims[0], (list, tuple)):
            dims = dims[0]
        return INDArray(self._t.permute(*dims), self._order)

    def permutei(self, *dims):
        self._t = self.permute(*dims)._t
        return self

    def transpose(self):
        if self._t.dim() <= 1:
            return INDArray(self._t.reshape(-1, 1) if self._t.dim() == 1 else self._t, self._order)
        return INDArray(self._t.permute(*reversed(range(self._t.dim()))), self._order)

    def transposei(self):
        self._t = self.transpose()._t
        return self

    def swapAxes(self, a, b):
        return INDArray(self._t.transpose(a, b), self._order)

    def broadcast(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
            shape = shape[0]
        return INDArray(self._t.expand(*shape), self._order)

    def repeat(self, dim, times):
        return INDArray(self._t.repeat_interleave(int(times), dim=dim), self._order)

    def repmat(self, *reps):
        return INDArray(self._t.repeat(*reps), self._order)

    # -- element access
    def _flat_index(self, i):
        return tuple(int(v) for v in np.unravel_index(int(i), tuple(self._t.shape), order="F" if self._order == "f" else "C"))

    def _key(self, idx):
        if len(idx) == 1 and self._t.dim() != 1:
            return self._flat_index(idx[0])
        return tuple(int(i) for i in idx)

    def getDouble(self, *idx):
        return float(self._t[self._key(idx)])

    getFloat = getDouble

    def getInt(self, *idx):
        return int(self._t[self._key(idx)])

    getLong = getInt

    def getNumber(self, *idx):
        return self._t[self._key(idx)].item()

    def getScalar(self, *idx):
        return INDArray(self._t[self._key(idx)].reshape(()), self._order)

    def putScalar(self, *args):
        *idx, v = args
        if len(idx) == 1 and isinstance(idx[0], (list, tuple)):
            idx = idx[0]
        self._t[self._key(idx)] = v
        return self

    def get(self, *indexes):
        return INDArray(self._t[NDArrayIndex._to_key(indexes, self._t.dim())], self._order)

    def put(self, indexes, value):
        if not isinstance(indexes, (list, tuple)):
            indexes = [indexes]
        v = _unwrap(value)
        self._t[NDArrayIndex._to_key(indexes, self._t.dim())] = v
        return self

    def getRow(self, i, keepDim=False):
        r = self._t[i]
        return INDArray(r.unsqueeze(0) if keepDim else r, self._order)

    def getColumn(self, i, keepDim=False):
        c = self._t[:, i]
        return INDArray(c.unsqueeze(1) if keepDim else c, self._order)

    def getRows(self, *rows):
        rows = rows[0] if len(rows) == 1 and isinstance(rows[0], (list, tuple)) else rows
        return INDArray(self._t[list(rows)], self._order)

    def getColumns(self, *cols):
        cols = cols[0] if len(cols) == 1 and isinstance(cols[0], (list, tuple)) else cols
        return INDArray(self._t[:, list(cols)], self._order)

    def putRow(self, i, row):
        self._t[i] = _unwrap(row).reshape(self._t[i].shape)
        return self

    def putColumn(self, i, col):
        self._t[:, i] = _unwrap(col).reshape(self._t[:, i].shape)
        return self

    def slice(self, i, dim=0):
        return INDArray(self._t.select(dim, i), self._order)

    def slices(self):
        return self._t.shape[0]

    def putSlice(self, i, put):
        self._t[i] = _unwrap(put)
        return self

    def tensorsAlongDimension(self, *dims):
        dims = [d % self._t.dim() for d in dims]
        return int(math.prod(self._t.shape[k] for k in range(self._t.dim()) if k not in dims))

    def tensorAlongDimension(self, index, *dims):
        """The ``index``-th sub-array spanning ``dims`` (remaining dims enumerated in c order) — a view."""
        r = self._t.dim()
        dims = sorted(d % r for d in dims)
        outer = [k for k in range(r) if k not in dims]
        oshape = [self._t.shape[k] for k in outer]
        coords = np.unravel_index(int(index), oshape) if oshape else ()
        key = [slice(None)] * r
        for k, c in zip(outer, coords):
            key[k] = int(c)
        return INDArray(self._t[tuple(key)], self._order)

    vectorAlongDimension = lambda self, index, dim: self.tensorAlongDimension(index, dim)

    def vectorsAlongDimension(self, dim):
        return self.tensorsAlongDimension(dim)

    # -- arithmetic (pairwise / scalar / broadcast through the legacy executioner)
    def _bin(self, name, other, inplace, reverse_scalar=None):
        o = _unwrap(other)
        x = self._t
        z = x if inplace else None
        if isinstance(o, torch.Tensor):
            o = o.to(x.device)
            if o.dtype != x.dtype:
                o = o.to(x.dtype)
            if inplace and torch.broadcast_shapes(x.shape, o.shape) != x.shape:
                raise ValueError("in-place op: operand of shape %s does not broadcast into %s" % (list(o.shape), list(x.shape)))
            r = _ex()._pairwise(name, x, o, z, None)
        else:
            r = _ex()._pairwise(name, x, None, z, None, scalar=float(o))
        return self if inplace else INDArray(r, self._order)

    def add(self, o): return self._bin("Add", o, False)
    def addi(self, o): return self._bin("Add", o, True)
    def sub(self, o): return self._bin("Subtract", o, False)
    def subi(self, o): return self._bin("Subtract", o, True)
    def mul(self, o): return self._bin("Multiply", o, False)
    def muli(self, o): return self._bin("Multiply", o, True)
    def div(self, o): return self._bin("Divide", o, False)
    def divi(self, o): return self._bin("Divide", o, True)
    def rsub(self, o): return self._bin("ReverseSubtract", o, False)
    def rsubi(self, o): return self._bin("ReverseSubtract", o, True)
    def rdiv(self, o): return self._bin("ReverseDivide", o, False)
    def rdivi(self, o): return self._bin("ReverseDivide", o, True)
    def fmod(self, o): return self._bin("FMod", o, False)
    def fmodi(self, o): return self._bin("FMod", o, True)
    def remainder(self, o): return self._bin("Remainder", o, False)

    def neg(self): return INDArray(_ex()._transform("Neg", self._t, None, None), self._order)

    def negi(self):
        _ex()._transform("Neg", self._t, self._t, None)
        return self

    __add__ = add
    __sub__ = sub
    __mul__ = mul
    __truediv__ = div
    __neg__ = neg
    __radd__ = add
    __rmul__ = mul
    __rsub__ = rsub
    __rtruediv__ = rdiv

    def _vec(self, name, v, dim, inplace):
        v = _unwrap(v).reshape(-1).to(self._t.device, self._t.dtype)
        shape = [1] * self._t.dim()
        shape[dim] = v.numel()
        return self._bin(name, v.reshape(shape), inplace)

    def addRowVector(self, v): return self._vec("Add", v, -1, False)
    def addiRowVector(self, v): return self._vec("Add", v, -1, True)
    def subRowVector(self, v): return self._vec("Subtract", v, -1, False)
    def subiRowVector(self, v): return self._vec("Subtract", v, -1, True)
    def mulRowVector(self, v): return self._vec("Multiply", v, -1, False)
    def muliRowVector(self, v): return self._vec("Multiply", v, -1, True)
    def divRowVector(self, v): return self._vec("Divide", v, -1, False)
    def diviRowVector(self, v): return self._vec("Divide", v, -1, True)
    def rsubRowVector(self, v): return self._vec("ReverseSubtract", v, -1, False)
    def rdivRowVector(self, v): return self._vec("ReverseDivide", v, -1, False)
    def addColumnVector(self, v): return self._vec("Add", v, 0, False)
    def addiColumnVector(self, v): return self._vec("Add", v, 0, True)
    def subColumnVector(self, v): return self._vec("Subtract", v, 0, False)
    def subiColumnVector(self, v): return self._vec("Subtract", v, 0, True)
    def mulColumnVector(self, v): return self._vec("Multiply", v, 0, False)
    def muliColumnVector(self, v): return self._vec("Multiply", v, 0, True)
    def divColumnVector(self, v): return self._vec("Divide", v, 0, False)
    def diviColumnVector(self, v): return self._vec("Divide", v, 0, True)

    # -- linear algebra
    def mmul(self, other, result=None, transposeA=False, transposeB=False):
        a, b = self._t, _unwrap(other)
        if a.dim() == 1:
            a = a.unsqueeze(0)
        if b.dim() == 1:
            b = b.unsqueeze(1) if a.shape[-1] == b.shape[0] and not transposeB else b.unsqueeze(0)
        if transposeA:
            a = a.transpose(-1, -2)
        if transposeB:
            b = b.transpose(-1, -2)
        r = torch.matmul(a, b.to(a.dtype))
        if result is not None:
            _unwrap(result).copy_(r)
            return result
        return INDArray(r, "f" if self._order == "f" else "c")

    def mmuli(self, other, result=None):
        return self.mmul(other, result if result is not None else None)

    __matmul__ = mmul

    # -- comparisons
    def _cmp(self, name, o):
        o = _unwrap(o)
        if isinstance(o, torch.Tensor):
            return INDArray(_ex()._pairwise(name, self._t, o.to(self._t.device, self._t.dtype), None, None, zbool=True), self._order)
        return INDArray(_ex()._pairwise(name, self._t, None, None, None, scalar=float(o), zbool=True), self._order)

    def gt(self, o): return self._cmp("GreaterThan", o)
    def gte(self, o): return self._cmp("GreaterThanOrEqual", o)
    def lt(self, o): return self._cmp("LessThan", o)
    def lte(self, o): return self._cmp("LessThanOrEqual", o)
    def eq(self, o): return self._cmp("EqualTo", o)
    def neq(self, o): return self._cmp("NotEqualTo", o)

    def equalsWithEps(self, other, eps=1e-5):
        o = _unwrap(other)
        if not isinstance(o, torch.Tensor) or list(o.shape) != list(self._t.shape):
            return False
        if self._t.numel() == 0:
            return True
        a, b = self._t.double(), o.to(self._t.device).double()
        return bool(((a - b).abs() <= eps).all()) if self._t.is_floating_point() else bool((a == b).all())

    def equals(self, other):
        return self.equalsWithEps(other, 1e-5)

    __eq__ = lambda self, o: self.equals(o) if isinstance(o, INDArray) else NotImplemented
    __hash__ = lambda self: id(self)

    def equalShapes(self, other):
        return list(self._t.shape) == list(_unwrap(other).shape)

    # -- reductions
    def _red(self, fam, name, dims, keepDims=False, extra=None):
        ex = _ex()
        dims = list(dims) if dims else None
        if len(dims or []) == 1 and isinstance(dims[0], (list, tuple)):
            dims = list(dims[0])
        zd = {"f": None}.get(fam)
        x = self._t
        if fam == "float":
            zdt = x.dtype if x.is_floating_point() else torch.float32
        elif fam == "same":
            zdt = x.dtype
        elif fam == "long":
            zdt = torch.int64
        else:
            zdt = torch.bool
        r = ex._reduce(name, x, dims, keepDims, extra, zdt)
        return INDArray(r, self._order)

    def sum(self, *dims, keepDims=False): return self._red("same", "Sum", dims, keepDims)
    def prod(self, *dims, keepDims=False): return self._red("same", "Prod", dims, keepDims)
    def max(self, *dims, keepDims=False): return self._red("same", "Max", dims, keepDims)
    def min(self, *dims, keepDims=False): return self._red("same", "Min", dims, keepDims)
    def amax(self, *dims): return self._red("same", "AMax", dims)
    def amin(self, *dims): return self._red("same", "AMin", dims)
    def mean(self, *dims, keepDims=False): return self._red("float", "Mean", dims, keepDims)
    def amean(self, *dims): return self._red("float", "AMean", dims)
    def norm1(self, *dims): return self._red("float", "Norm1", dims)
    def norm2(self, *dims): return self._red("float", "Norm2", dims)
    def normmax(self, *dims): return self._red("float", "NormMax", dims)
    def entropy(self, *dims): return self._red("float", "Entropy", dims)
    def shannonEntropy(self, *dims): return self._red("float", "ShannonEntropy", dims)
    def logEntropy(self, *dims): return self._red("float", "LogEntropy", dims)
    def any(self): return bool(self._red("bool", "Any", None)._t)
    def all(self): return bool(self._red("bool", "All", None)._t)
    def none(self): return not self.any()

    def var(self, *dims, biasCorrected=True):
        return INDArray(_ex().execSummaryStats(0, self._t, list(dims) or None, biasCorrected), self._order)

    def std(self, *dims, biasCorrected=True):
        return INDArray(_ex().execSummaryStats(1, self._t, list(dims) or None, biasCorrected), self._order)

    def argMax(self, *dims):
        return INDArray(_ex().execIndexReduce(0, self._t, list(dims) or None), self._order)

    def argMin(self, *dims):
        return INDArray(_ex().execIndexReduce(1, self._t, list(dims) or None), self._order)

    def sumNumber(self): return self.sum()._t.item()
    def meanNumber(self): return self.mean()._t.item()
    def maxNumber(self): return self.max()._t.item()
    def minNumber(self): return self.min()._t.item()
    def prodNumber(self): return self.prod()._t.item()
    def norm1Number(self): return self.norm1()._t.item()
    def norm2Number(self): return self.norm2()._t.item()
    def normmaxNumber(self): return self.normmax()._t.item()
    def stdNumber(self, biasCorrected=True): return self.std(biasCorrected=biasCorrected)._t.item()
    def varNumber(self, biasCorrected=True): return self.var(biasCorrected=biasCorrected)._t.item()
    def ameanNumber(self): return self.amean()._t.item()
    def amaxNumber(self): return self.amax()._t.item()
    def aminNumber(self): return self.amin()._t.item()

    def cumsum(self, dim):
        return INDArray(torch.cumsum(self._t, dim), self._order)

    def cumsumi(self, dim):
        self._t.copy_(torch.cumsum(self._t, dim))
        return self

    def distance1(self, o): return float(_ex().execReduce3(0, self._t, _unwrap(o)))
    def distance2(self, o): return float(_ex().execReduce3(1, self._t, _unwrap(o)))
    def squaredDistance(self, o): return self.distance2(o) ** 2

    # -- conversions
    def toDoubleVector(self): return self._t.detach().reshape(-1).double().cpu().tolist()
    def toFloatVector(self): return self._t.detach().reshape(-1).float().cpu().tolist()
    def toIntVector(self): return self._t.detach().reshape(-1).int().cpu().tolist()
    def toLongVector(self): return self._t.detach().reshape(-1).long().cpu().tolist()
    def toDoubleMatrix(self): return self._t.detach().double().cpu().tolist()
    def toFloatMatrix(self): return self._t.detach().float().cpu().tolist()
    def toIntMatrix(self): return self._t.detach().int().cpu().tolist()

    # -- remaining INDArray.java surface (conditions, statistics, vector-broadcast variants, descriptors, scope / compression flags)
    def _cond_mask(self, cond):
        """bool mask of a Condition (nd4j/common.py Conditions.*) or a python predicate over the tensor."""
        if hasattr(cond, "mask"):
            return cond.mask(self._t)
        return cond(self._t)

    def cond(self, condition):
        """1 where the condition holds, else 0 (same data type)."""
        return INDArray(self._cond_mask(condition).to(self._t.dtype), self._order)

    def match(self, comp, condition):
        """BOOL array: where does ``self <condition> comp`` hold (comp: number or array; condition gives the comparison)."""
        c = _unwrap(comp)
        name = type(condition).__name__.lower() if not isinstance(condition, str) else condition.lower()
        ops = {"equals": torch.eq, "epsequals": lambda a, b: (a - b).abs() < 1e-5, "notequals": torch.ne, "epsnotequals": lambda a, b: (a - b).abs() >= 1e-5,
               "greaterthan": torch.gt, "greaterthanorequal": torch.ge, "lessthan": torch.lt, "lessthanorequal": torch.le}
        key = [k for k in ops if name.replace("_", "").startswith(k)]
        if not key:
            return INDArray(self._cond_mask(condition), self._order)
        c = torch.as_tensor(c, device=self._t.device, dtype=self._t.dtype)
        return INDArray(ops[max(key, key=len)](self._t, c), self._order)

    def getWhere(self, comp, condition):
        """1-D array of the elements for which ``self <condition> comp`` holds (Nd4j getWhere / boolean_mask)."""
        m = self.match(comp, condition)._t if comp is not None else self._cond_mask(condition)
        return INDArray(self._t[m], self._order)

    def putWhere(self, comp, put, condition):
        """Copy of this array with ``put`` written where ``self <condition> comp`` holds."""
        m = self.match(comp, condition)._t
        out = self._t.clone()
        pv = _unwrap(put)
        out[m] = pv[m] if isinstance(pv, torch.Tensor) and pv.shape == out.shape else torch.as_tensor(pv, device=out.device, dtype=out.dtype)
        return INDArray(out, self._order)

    def putWhereWithMask(self, mask, put):
        m = _unwrap(mask).bool()
        out = self._t.clone()
        pv = _unwrap(put)
        out[m] = pv[m] if isinstance(pv, torch.Tensor) and pv.shape == out.shape else torch.as_tensor(pv, device=out.device, dtype=out.dtype)
        return INDArray(out, self._order)

    def replaceWhere(self, arr, condition):
        """In place: elements that satisfy the condition are replaced by the corresponding elements of ``arr``."""
        m = self._cond_mask(condition)
        self._t[m] = _unwrap(arr).to(self._t.dtype)[m]
        return self

    def assignIf(self, arr, condition):
        """In place: take arr's element wherever arr's element satisfies the condition (CompareAndSet semantics)."""
        a = _unwrap(arr).to(self._t.device, self._t.dtype)
        m = condition.mask(a) if hasattr(condition, "mask") else condition(a)
        self._t[m] = a[m]
        return self

    def scan(self, condition):
        """Number of elements that satisfy the condition (MatchCondition reduction)."""
        return int(self._cond_mask(condition).sum())

    def eps(self, other):
        o = torch.as_tensor(_unwrap(other), device=self._t.device, dtype=self._t.dtype)
        return INDArray((self._t - o).abs() < 1e-5, self._order)

    def isInfinite(self):
        return INDArray(torch.isinf(self._t), self._order)

    def isNaN(self):
        return INDArray(torch.isnan(self._t), self._order)

    def entropyNumber(self): return self.entropy()._t.item()
    def shannonEntropyNumber(self): return self.shannonEntropy()._t.item()
    def logEntropyNumber(self): return self.logEntropy()._t.item()

    def median(self, *dims):
        x = self._t if self._t.is_floating_point() else self._t.double()
        if not dims:
            return INDArray(torch.quantile(x.reshape(-1).double(), 0.5).to(x.dtype), self._order)
        return self.percentile(50.0, *dims)

    def medianNumber(self):
        return self.median()._t.item()

    def percentile(self, q, *dims):
        """Linear-interpolated percentile (q in 0..100) over the given dimensions (Percentile op / ops/declarable percentile)."""
        x = (self._t if self._t.is_floating_point() else self._t.double())
        if not dims:
            return INDArray(torch.quantile(x.reshape(-1).double(), float(q) / 100.0).to(x.dtype), self._order)
        dims = sorted(d % x.dim() for d in dims)
        keep = [d for d in range(x.dim()) if d not in dims]
        y = x.permute(*keep, *dims).reshape([x.shape[d] for d in keep] + [-1])
        return INDArray(torch.quantile(y.double(), float(q) / 100.0, dim=-1).to(x.dtype), self._order)

    def percentileNumber(self, q):
        return self.percentile(q)._t.item()

    def remainderi(self, other):
        self._t.copy_(torch.remainder(self._t, torch.as_tensor(_unwrap(other), device=self._t.device, dtype=self._t.dtype)))
        return self

    def rsubiRowVector(self, v): return self._vec("ReverseSubtract", v, -1, True)
    def rdiviRowVector(self, v): return self._vec("ReverseDivide", v, -1, True)
    def rsubColumnVector(self, v): return self._vec("ReverseSubtract", v, 0, False)
    def rsubiColumnVector(self, v): return self._vec("ReverseSubtract", v, 0, True)
    def rdivColumnVector(self, v): return self._vec("ReverseDivide", v, 0, False)
    def rdiviColumnVector(self, v): return self._vec("ReverseDivide", v, 0, True)

    def putiRowVector(self, v):
        self._t.copy_(_unwrap(v).reshape(1, -1).to(self._t.device, self._t.dtype).expand_as(self._t))
        return self

    def putiColumnVector(self, v):
        self._t.copy_(_unwrap(v).reshape(-1, 1).to(self._t.device, self._t.dtype).expand_as(self._t))
        return self

    def isVectorOrScalar(self): return self.isVector() or self._t.numel() == 1
    def isRowVectorOrScalar(self): return self.isRowVector() or self._t.numel() == 1
    def isColumnVectorOrScalar(self): return (self._t.dim() == 2 and self._t.shape[1] == 1) or self._t.numel() == 1
    def isS(self): return False                 # no UTF string arrays on the device
    def isSparse(self): return False
    def isCompressed(self): return bool(getattr(self, "_compressed", False))

    def markAsCompressed(self, reallyCompressed):
        self._compressed = bool(reallyCompressed)

    def isInScope(self):
        """An array attached to a workspace is usable only while that workspace's generation is alive."""
        ws = getattr(self, "_ws", None)
        if ws is None:
            return True
        from .workspace import getWorkspaceManager
        return getWorkspaceManager().checkIfWorkspaceExists(ws)

    def leverageOrDetach(self, workspaceId):
        from .workspace import getWorkspaceManager
        if getWorkspaceManager().checkIfWorkspaceExistsAndActive(workspaceId):
            return self.leverageTo(workspaceId)
        return self.detach()

    def getId(self):
        return id(self)

    def setCloseable(self, flag):
        self._closeable = bool(flag)

    def element(self):
        if self._t.numel() != 1:
            raise ValueError("Unable to retrieve element from non scalars")
        return self._t.item()

    def getDoubleUnsafe(self, offset):
        return float(self.data()[int(offset)]) if not isinstance(self.data(), torch.Tensor) else float(self._t.reshape(-1)[int(offset)])

    def putScalarUnsafe(self, offset, value):
        self._t.view(-1)[int(offset)] = value
        return self

    def getString(self, index):
        raise TypeError("getString: this array is not of a UTF8 data type")

    def linearIndex(self, i):
        """Buffer offset of the i-th element in this array's iteration order."""
        t = self._t
        idx, off = int(i), 0
        dims = range(t.dim() - 1, -1, -1) if self._order == "c" else range(t.dim())
        for d in dims:
            off += (idx % t.shape[d]) * t.stride(d)
            idx //= t.shape[d]
        return off + t.storage_offset()

    def originalOffset(self):
        return self._t.storage_offset()

    def getLeadingOnes(self):
        n = 0
        for s_ in self._t.shape:
            if s_ != 1:
                break
            n += 1
        return n

    def getTrailingOnes(self):
        n = 0
        for s_ in reversed(self._t.shape):
            if s_ != 1:
                break
            n += 1
        return n

    def dimShuffle(self, rearrange, newOrder, broadCastable):
        """Theano-style dimshuffle: ``rearrange`` entries are dimension indices or 'x' (new broadcast axis of size 1)."""
        t = self._t
        used = [r for r in rearrange if r != "x"]
        dropped = [d for d in range(t.dim()) if d not in used]
        if any(t.shape[d] != 1 for d in dropped):
            raise ValueError("dimShuffle can only drop dimensions of size 1")
        t = t.permute(*used, *dropped).reshape([t.shape[d] for d in used])
        for pos, r in enumerate(rearrange):
            if r == "x":
                t = t.unsqueeze(pos)
        return INDArray(t, self._order)

    def sliceVectors(self, out=None):
        out = [] if out is None else out
        if self._t.dim() <= 1:
            out.append(self)
        else:
            for k in range(self._t.shape[0]):
                INDArray(self._t[k], self._order).sliceVectors(out)
        return out

    def setData(self, data):
        src = data.tensor() if hasattr(data, "tensor") else torch.as_tensor(_unwrap(data))
        self._t.view(-1).copy_(src.reshape(-1).to(self._t.device, self._t.dtype))

    def setOrder(self, order):
        assert order in ("c", "f")
        self._order = order

    def setShapeAndStride(self, shape, stride):
        self._t = torch.as_strided(self._t, [int(v) for v in shape], [int(v) for v in stride], self._t.storage_offset())

    def shapeInfoJava(self):
        return list(self.shapeInfo())

    def shapeInfoDataBuffer(self):
        from .common import DataBuffer
        return DataBuffer(torch.tensor(list(self.shapeInfo()), dtype=torch.int64))

    def shapeDescriptor(self):
        from .common import LongShapeDescriptor
        return LongShapeDescriptor.fromShape(list(self._t.shape), self.dataType()) if hasattr(LongShapeDescriptor, "fromShape") else \
            LongShapeDescriptor(list(self._t.shape), list(self._t.stride()), self._t.storage_offset(), 1, self._order, self.dataType())

    def toFlatArray(self, builder=None):
        """FlatBuffers FlatArray bytes of this array (graph serialisation wire format, autodiff/serde.py)."""
        from ..autodiff import serde as ASD
        return ASD.flat_array_bytes(self._t) if hasattr(ASD, "flat_array_bytes") else serde.to_bytes(self._t)

    def toLongMatrix(self): return self._t.detach().long().cpu().tolist()

    def toStringFull(self):
        with torch._tensor_str.printoptions(threshold=1 << 62, precision=8):
            return repr(self._t)

    def unsafeDuplication(self, blocking=True):
        out = INDArray(self._t.clone(), self._order)
        if blocking and self._t.is_cuda:
            torch.cuda.current_stream().synchronize()
        return out

    def __len__(self):
        return self._t.shape[0] if self._t.dim() else 1

    def __getitem__(self, k):
        return INDArray(self._t[k], self._order)

    def __setitem__(self, k, v):
        self._t[k] = _unwrap(v)

    def __repr__(self):
        return "INDArray(%s, order=%s)\n%s" % (self.dataType(), self._order, self._t)

    toString = __repr__
