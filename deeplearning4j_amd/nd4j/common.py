"""Small ND4J API objects that user code touches directly: shape utilities, conditions, data buffers, preconditions,
resources. Reference packages: nd4j-api org/nd4j/linalg/api/shape (Shape.java, LongShapeDescriptor.java),
org/nd4j/linalg/indexing/conditions (Conditions.java, ConditionBuilder), org/nd4j/linalg/api/ops/impl/reduce/longer/
MatchCondition.java, org/nd4j/linalg/api/buffer (DataBuffer.java, BaseDataBuffer.java), nd4j-common
(org/nd4j/common/base/Preconditions.java, primitives/Pair.java, io/ClassPathResource.java, util/ArchiveUtils.java,
util/ArrayUtil.java, resources/Resources.java).
"""
from __future__ import annotations

import gzip
import os
import shutil
import tarfile
import zipfile
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch

from .compression import CompressedArray
from .rng import Random
from .workspace import MemoryWorkspace


# ---------------------------------------------------------------------------------------------- shapes
class Shape:
    """Static shape helpers (rank/stride/broadcast arithmetic on plain int sequences)."""

    @staticmethod
    def length(shape: Sequence[int]) -> int:
        n = 1
        for s in shape:
            n *= int(s)
        return n

    prodLong = length

    @staticmethod
    def calcStrides(shape: Sequence[int], order: str = "c") -> List[int]:
        st, acc = [0] * len(shape), 1
        idx = range(len(shape) - 1, -1, -1) if order == "c" else range(len(shape))
        for i in idx:
            st[i] = acc
            acc *= max(int(shape[i]), 1)
        return st

    @staticmethod
    def getOffset(baseOffset: int, shape: Sequence[int], strides: Sequence[int], *indices) -> int:
        if len(indices) == 1 and isinstance(indices[0], (list, tuple)):
            indices = tuple(indices[0])
        off = baseOffset
        for i, s, st in zip(indices, shape, strides):
            if not 0 <= i < s:
                raise IndexError("index %d out of bounds for dimension of size %d" % (i, s))
            off += i * st
        return off

    @staticmethod
    def ind2sub(shape: Sequence[int], index: int, order: str = "c") -> List[int]:
        out = [0] * len(shape)
        idx = range(len(shape) - 1, -1, -1) if order == "c" else range(len(shape))
        for i in idx:
            out[i] = index % shape[i]
            index //= shape[i]
        return out

    @staticmethod
    def sub2Ind(shape: Sequence[int], indices: Sequence[int], order: str = "c") -> int:
        return Shape.getOffset(0, shape, Shape.calcStrides(shape, order), *indices)

    @staticmethod
    def isVector(shape) -> bool:
        return len(shape) == 1 or (len(shape) == 2 and (shape[0] == 1 or shape[1] == 1))

    @staticmethod
    def isRowVectorShape(shape) -> bool:
        return len(shape) == 1 or (len(shape) == 2 and shape[0] == 1)

    @staticmethod
    def isColumnVectorShape(shape) -> bool:
        return len(shape) == 2 and shape[1] == 1

    @staticmethod
    def isMatrix(shape) -> bool:
        return len(shape) == 2 and shape[0] != 1 and shape[1] != 1

    @staticmethod
    def isScalar(shape) -> bool:
        return len(shape) == 0 or Shape.length(shape) == 1

    @staticmethod
    def squeeze(shape) -> List[int]:
        return [s for s in shape if s != 1]

    @staticmethod
    def shapeEquals(a, b) -> bool:
        return list(a) == list(b) or (Shape.isVector(a) and Shape.isVector(b) and Shape.length(a) == Shape.length(b)
                                      and len(Shape.squeeze(a)) <= 1 and len(Shape.squeeze(b)) <= 1)

    @staticmethod
    def broadcastOutputShape(a: Sequence[int], b: Sequence[int]) -> List[int]:
        return list(torch.broadcast_shapes(tuple(a), tuple(b)))

    @staticmethod
    def areShapesBroadcastable(a, b) -> bool:
        try:
            torch.broadcast_shapes(tuple(a), tuple(b))
            return True
        except RuntimeError:
            return False

    @staticmethod
    def normalizeAxis(rank: int, *axis) -> List[int]:
        if len(axis) == 1 and isinstance(axis[0], (list, tuple)):
            axis = tuple(axis[0])
        if not axis or (len(axis) == 1 and axis[0] == 2 ** 31 - 1):
            return list(range(rank))
        out = sorted({a + rank if a < 0 else a for a in axis})
        if out and not 0 <= out[0] <= out[-1] < rank:
            raise ValueError("axis %s out of range for rank %d" % (list(axis), rank))
        return out

    @staticmethod
    def getReducedShape(shape: Sequence[int], dimensions: Sequence[int], keepDims: bool = False) -> List[int]:
        dims = set(Shape.normalizeAxis(len(shape), *dimensions))
        if keepDims:
            return [1 if i in dims else s for i, s in enumerate(shape)]
        return [s for i, s in enumerate(shape) if i not in dims]

    @staticmethod
    def getOrder(shape: Sequence[int], strides: Sequence[int], elementStride: int = 1) -> str:
        c = list(strides) == [s * elementStride for s in Shape.calcStrides(shape, "c")]
        f = list(strides) == [s * elementStride for s in Shape.calcStrides(shape, "f")]
        return "a" if c and f else "c" if c else "f" if f else "c"

    @staticmethod
    def pickPairwiseDataType(a: torch.dtype, b: torch.dtype) -> torch.dtype:
        return torch.promote_types(a, b)


class LongShapeDescriptor:
    """(shape, strides, element-wise stride, order, dtype) with value equality — the key the reference's shape cache uses
    (ConstantShapeHelper, LongShapeDescriptor.java)."""

    __slots__ = ("shape", "stride", "ews", "order", "dtype", "empty")

    def __init__(self, shape, stride=None, ews: int = 1, order: str = "c", dtype: torch.dtype = torch.float32, empty=False):
        self.shape = tuple(int(s) for s in shape)
        self.stride = tuple(stride) if stride is not None else tuple(Shape.calcStrides(self.shape, order))
        self.ews, self.order, self.dtype, self.empty = ews, order, dtype, bool(empty)

    @staticmethod
    def fromShape(shape, dtype=torch.float32, order="c"):
        return LongShapeDescriptor(shape, None, 1, order, dtype)

    @staticmethod
    def fromTensor(t: torch.Tensor):
        order = Shape.getOrder(t.shape, t.stride())
        return LongShapeDescriptor(t.shape, t.stride(), 1 if t.is_contiguous() else 0, order, t.dtype, t.numel() == 0)

    def rank(self):
        return len(self.shape)

    def length(self):
        return 0 if self.empty else Shape.length(self.shape)

    def asDataType(self, dtype):
        return LongShapeDescriptor(self.shape, self.stride, self.ews, self.order, dtype, self.empty)

    def toShapeInfo(self):
        from .serde import shape_info
        names = {torch.float32: "FLOAT", torch.float64: "DOUBLE", torch.float16: "HALF", torch.bfloat16: "BFLOAT16",
                 torch.int64: "LONG", torch.int32: "INT", torch.int16: "SHORT", torch.int8: "BYTE"}
        return shape_info(list(self.shape), self.order, names.get(self.dtype, "FLOAT"), list(self.stride), self.empty)

    def _k(self):
        return (self.shape, self.stride, self.ews, self.order, self.dtype, self.empty)

    def __eq__(self, o):
        return isinstance(o, LongShapeDescriptor) and self._k() == o._k()

    def __hash__(self):
        return hash(self._k())

    def __repr__(self):
        return "LongShapeDescriptor(shape=%s, stride=%s, order=%s, dtype=%s)" % (list(self.shape), list(self.stride), self.order, self.dtype)


class ArrayUtil:
    @staticmethod
    def prod(xs) -> int:
        return Shape.length(xs)

    @staticmethod
    def flatten(nested) -> list:
        out = []
        for x in nested:
            out.extend(ArrayUtil.flatten(x) if isinstance(x, (list, tuple)) else [x])
        return out

    @staticmethod
    def range(a, b, step=1) -> List[int]:
        return list(range(a, b, step))

    @staticmethod
    def removeIndex(xs, *idx) -> list:
        drop = set(i % len(xs) for i in idx)
        return [x for i, x in enumerate(xs) if i not in drop]

    @staticmethod
    def reverseCopy(xs) -> list:
        return list(xs)[::-1]

    @staticmethod
    def argsort(xs, ascending=True) -> List[int]:
        return sorted(range(len(xs)), key=lambda i: xs[i], reverse=not ascending)

    @staticmethod
    def toLongArray(xs) -> List[int]:
        return [int(x) for x in xs]

    @staticmethod
    def calcStrides(shape):
        return Shape.calcStrides(shape, "c")

    @staticmethod
    def calcStridesFortran(shape):
        return Shape.calcStrides(shape, "f")


# ---------------------------------------------------------------------------------------------- conditions
class Condition:
    """Elementwise predicate with a scalar operand; ``mask(x)`` is the boolean tensor, ``apply(v)`` the scalar test."""

    def __init__(self, name: str, fn: Callable, value=0.0, eps: float = 0.0):
        self.name, self._fn, self.value, self.eps = name, fn, value, eps

    def mask(self, x: torch.Tensor) -> torch.Tensor:
        return self._fn(x, self.value, self.eps)

    def apply(self, v) -> bool:
        return bool(self.mask(torch.as_tensor(v, dtype=torch.float64)))

    def getValue(self):
        return self.value

    def epsilon(self):
        return self.eps

    def And(self, other: "Condition"):
        return Condition("and", lambda x, *_: self.mask(x) & other.mask(x))

    def Or(self, other: "Condition"):
        return Condition("or", lambda x, *_: self.mask(x) | other.mask(x))

    def Not(self):
        return Condition("not", lambda x, *_: ~self.mask(x))

    def __repr__(self):
        return "Condition(%s, %s)" % (self.name, self.value)


class Conditions:
    """Factory of the reference's condition set (Conditions.java)."""

    @staticmethod
    def equals(v=0.0):
        return Condition("equals", lambda x, v, e: x == v, v)

    @staticmethod
    def notEquals(v=0.0):
        return Condition("notEquals", lambda x, v, e: x != v, v)

    @staticmethod
    def epsEquals(v=0.0, eps=1e-5):
        return Condition("epsEquals", lambda x, v, e: (x - v).abs() <= e, v, eps)

    @staticmethod
    def epsNotEquals(v=0.0, eps=1e-5):
        return Condition("epsNotEquals", lambda x, v, e: (x - v).abs() > e, v, eps)

    @staticmethod
    def greaterThan(v=0.0):
        return Condition("greaterThan", lambda x, v, e: x > v, v)

    @staticmethod
    def greaterThanOrEqual(v=0.0):
        return Condition("greaterThanOrEqual", lambda x, v, e: x >= v, v)

    @staticmethod
    def lessThan(v=0.0):
        return Condition("lessThan", lambda x, v, e: x < v, v)

    @staticmethod
    def lessThanOrEqual(v=0.0):
        return Condition("lessThanOrEqual", lambda x, v, e: x <= v, v)

    @staticmethod
    def absGreaterThan(v=0.0):
        return Condition("absGreaterThan", lambda x, v, e: x.abs() > v, v)

    @staticmethod
    def absGreaterThanOrEqual(v=0.0):
        return Condition("absGreaterThanOrEqual", lambda x, v, e: x.abs() >= v, v)

    @staticmethod
    def absLessThan(v=0.0):
        return Condition("absLessThan", lambda x, v, e: x.abs() < v, v)

    @staticmethod
    def absLessThanOrEqual(v=0.0):
        return Condition("absLessThanOrEqual", lambda x, v, e: x.abs() <= v, v)

    @staticmethod
    def isNan():
        return Condition("isNan", lambda x, v, e: torch.isnan(x) if x.is_floating_point() else torch.zeros_like(x, dtype=torch.bool))

    @staticmethod
    def isInfinite():
        return Condition("isInfinite", lambda x, v, e: torch.isinf(x) if x.is_floating_point() else torch.zeros_like(x, dtype=torch.bool))

    @staticmethod
    def isFinite():
        return Condition("isFinite", lambda x, v, e: torch.isfinite(x) if x.is_floating_point() else torch.ones_like(x, dtype=torch.bool))

    @staticmethod
    def notFinite():
        return Conditions.isFinite().Not()


def _raw(x):
    return x.tensor if hasattr(x, "tensor") and isinstance(getattr(x, "tensor"), torch.Tensor) else torch.as_tensor(x)


class MatchCondition:
    """Count of elements satisfying a condition, over the whole array or along dimensions (int64 result)."""

    def __init__(self, x, condition: Condition, *dimensions, keepDims: bool = False):
        self.x, self.condition, self.dimensions, self.keepDims = x, condition, dimensions, keepDims

    def exec(self) -> torch.Tensor:
        m = self.condition.mask(_raw(self.x))
        if not self.dimensions:
            return m.sum(dtype=torch.int64)
        return m.sum(dim=tuple(self.dimensions), keepdim=self.keepDims, dtype=torch.int64)


class BooleanIndexing:
    """In-place conditional replacement and predicates (reference: indexing/BooleanIndexing.java)."""

    @staticmethod
    def replaceWhere(to, setOrFrom, condition: Condition):
        t = _raw(to)
        m = condition.mask(t)
        if isinstance(setOrFrom, (int, float)):
            t.masked_fill_(m, setOrFrom)
        else:
            src = _raw(setOrFrom).to(t.dtype)
            t.copy_(torch.where(m, src.expand_as(t), t))
        return to

    @staticmethod
    def and_(x, condition: Condition) -> bool:
        return bool(condition.mask(_raw(x)).all())

    @staticmethod
    def or_(x, condition: Condition) -> bool:
        return bool(condition.mask(_raw(x)).any())

    @staticmethod
    def firstIndex(x, condition: Condition) -> int:
        nz = condition.mask(_raw(x)).reshape(-1).nonzero()
        return int(nz[0]) if nz.numel() else -1

    @staticmethod
    def lastIndex(x, condition: Condition) -> int:
        nz = condition.mask(_raw(x)).reshape(-1).nonzero()
        return int(nz[-1]) if nz.numel() else -1

    @staticmethod
    def chooseFrom(x, condition: Condition):
        t = _raw(x)
        return t[condition.mask(t)]


# ---------------------------------------------------------------------------------------------- buffers
class DataBuffer:
    """Linear typed view of an array's storage (reference: DataBuffer.java). Wraps a 1-D tensor that aliases the array."""

    def __init__(self, data, dtype: Optional[torch.dtype] = None, device=None):
        if isinstance(data, int):
            self._t = torch.zeros(data, dtype=dtype or torch.float32, device=device)
        else:
            t = _raw(data)
            self._t = (t if dtype is None else t.to(dtype)).reshape(-1) if t.is_contiguous() else t.contiguous().reshape(-1)

    def length(self) -> int:
        return self._t.numel()

    def dataType(self) -> torch.dtype:
        return self._t.dtype

    def getElementSize(self) -> int:
        return self._t.element_size()

    def asTensor(self) -> torch.Tensor:
        return self._t

    def getDouble(self, i) -> float:
        return float(self._t[i])

    getFloat = getNumber = getDouble

    def getInt(self, i) -> int:
        return int(self._t[i])

    getLong = getInt

    def put(self, i, v):
        self._t[i] = v

    def assign(self, v):
        if isinstance(v, DataBuffer):
            self._t.copy_(v._t)
        else:
            self._t.fill_(v)
        return self

    def asFloat(self):
        return self._t.to(torch.float32).tolist()

    asDouble = asFloat

    def asInt(self):
        return self._t.to(torch.int64).tolist()

    asLong = asInt

    def asBytes(self) -> bytes:
        return self._t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()

    def dup(self):
        return DataBuffer(self._t.clone())

    def address(self) -> int:
        return self._t.data_ptr()

    def sameUnderlyingData(self, other: "DataBuffer") -> bool:
        return self._t.untyped_storage().data_ptr() == other._t.untyped_storage().data_ptr()

    def isConstant(self):
        return False

    def __len__(self):
        return self.length()


class CompressedDataBuffer(DataBuffer):
    """Opaque payload of a compressed array plus its descriptor (reference: compression/CompressedDataBuffer.java)."""

    def __init__(self, compressed: CompressedArray):
        payload = compressed.payload if isinstance(compressed.payload, torch.Tensor) else torch.as_tensor(compressed.payload)
        super().__init__(payload)
        self.compressed = compressed

    def getCompressionDescriptor(self):
        c = self.compressed
        n = 1
        for s in c.shape:
            n *= s
        return {"compressionAlgorithm": c.algorithm, "originalLength": n * torch.empty((), dtype=c.dtype).element_size(),
                "compressedLength": self.length() * self.getElementSize(), "originalElementSize": torch.empty((), dtype=c.dtype).element_size(),
                "numberOfElements": n}


# ---------------------------------------------------------------------------------------------- aliases onto the runtime
class NativeRandom(Random):
    """The counter-based generator of rng.py under the reference's class name (nd4j-native rng/NativeRandom.java)."""


class Nd4jWorkspace(MemoryWorkspace):
    """The arena workspace of workspace.py under the reference's class name (memory/abstracts/Nd4jWorkspace.java)."""


# ---------------------------------------------------------------------------------------------- nd4j-common
class Preconditions:
    """Argument/state checks with %s formatting (nd4j-common base/Preconditions.java)."""

    @staticmethod
    def _fmt(msg, args):
        if not args:
            return msg
        try:
            return msg % tuple(args)
        except TypeError:
            return "%s %s" % (msg, list(args))

    @staticmethod
    def checkArgument(cond: bool, msg: str = "", *args):
        if not cond:
            raise ValueError(Preconditions._fmt(msg, args))

    @staticmethod
    def checkState(cond: bool, msg: str = "", *args):
        if not cond:
            raise RuntimeError(Preconditions._fmt(msg, args))

    @staticmethod
    def checkNotNull(o, msg: str = "", *args):
        if o is None:
            raise TypeError(Preconditions._fmt(msg or "null argument", args))
        return o

    @staticmethod
    def throwEx(msg: str, *args):
        raise ValueError(Preconditions._fmt(msg, args))


class Pair(tuple):
    def __new__(cls, first=None, second=None):
        return super().__new__(cls, (first, second))

    @staticmethod
    def of(a, b):
        return Pair(a, b)

    makePair = of

    def getFirst(self):
        return self[0]

    getKey = getLeft = getFirst

    def getSecond(self):
        return self[1]

    getValue = getRight = getSecond


class Triple(tuple):
    def __new__(cls, a=None, b=None, c=None):
        return super().__new__(cls, (a, b, c))

    def getFirst(self):
        return self[0]

    def getSecond(self):
        return self[1]

    def getThird(self):
        return self[2]


class ClassPathResource:
    """A file looked up on a search path: the directories of ``DL4J_AMD_RESOURCES`` (os.pathsep-separated), then the
    package directory, then the working directory."""

    def __init__(self, path: str):
        self.path = path

    def _roots(self):
        env = os.environ.get("DL4J_AMD_RESOURCES", "")
        pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        return [p for p in env.split(os.pathsep) if p] + [pkg, os.getcwd()]

    def exists(self) -> bool:
        return self._find() is not None

    def _find(self):
        if os.path.isabs(self.path):
            return self.path if os.path.exists(self.path) else None
        for r in self._roots():
            p = os.path.join(r, self.path)
            if os.path.exists(p):
                return p
        return None

    def getFile(self) -> str:
        p = self._find()
        if p is None:
            raise FileNotFoundError("resource '%s' not found under %s" % (self.path, self._roots()))
        return p

    def getInputStream(self):
        return open(self.getFile(), "rb")

    def getFilename(self):
        return os.path.basename(self.path)


class Resources:
    @staticmethod
    def exists(path) -> bool:
        return ClassPathResource(path).exists()

    @staticmethod
    def asFile(path) -> str:
        return ClassPathResource(path).getFile()

    @staticmethod
    def asStream(path):
        return ClassPathResource(path).getInputStream()


class ArchiveUtils:
    """zip / tar / tar.gz / gz extraction with path-traversal protection (nd4j-common util/ArchiveUtils.java)."""

    @staticmethod
    def _safe(dest, name):
        target = os.path.realpath(os.path.join(dest, name))
        root = os.path.realpath(dest)
        if target != root and not target.startswith(root + os.sep):
            raise ValueError("archive entry '%s' escapes the destination directory" % name)
        return target

    @staticmethod
    def unzipFileTo(file: str, dest: str, logFiles: bool = False):
        os.makedirs(dest, exist_ok=True)
        low = file.lower()
        if low.endswith(".zip") or low.endswith(".jar"):
            with zipfile.ZipFile(file) as z:
                for n in z.namelist():
                    ArchiveUtils._safe(dest, n)
                z.extractall(dest)
        elif low.endswith((".tar.gz", ".tgz", ".tar")):
            with tarfile.open(file) as t:
                members = t.getmembers()
                for m in members:
                    ArchiveUtils._safe(dest, m.name)
                    if m.issym() or m.islnk():
                        raise ValueError("archive entry '%s' is a link" % m.name)
                t.extractall(dest, members=members)
        elif low.endswith(".gz"):
            out = os.path.join(dest, os.path.basename(file)[:-3])
            with gzip.open(file, "rb") as src, open(out, "wb") as dst:
                shutil.copyfileobj(src, dst)
        else:
            raise ValueError("unknown archive type: %s" % file)

    @staticmethod
    def zipListFiles(file: str) -> List[str]:
        with zipfile.ZipFile(file) as z:
            return [n for n in z.namelist() if not n.endswith("/")]

    @staticmethod
    def tarListFiles(file: str) -> List[str]:
        with tarfile.open(file) as t:
            return [m.name for m in t.getmembers() if m.isfile()]

    @staticmethod
    def zipExtractSingleFile(zipFile: str, destination: str, pathInZip: str):
        with zipfile.ZipFile(zipFile) as z, z.open(pathInZip) as src, open(destination, "wb") as dst:
            shutil.copyfileobj(src, dst)
