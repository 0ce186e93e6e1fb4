"""Aggregate ops: many small independent operations executed as one batch (one kernel launch on the GPU).

Reference: nd4j-api/.../linalg/api/ops/aggregates/{Aggregate,BaseAggregate,Batch}.java, impl/{AggregateAxpy,AggregateGEMM}.java (plus
the Dot / HierarchicSoftmax / SkipGram / CBOW aggregates of older releases, opNums libnd4j/include/loops/legacy_ops.h:26-28), executed
by OpExecutioner.exec(Aggregate) / exec(Batch) -> NativeOps::execAggregate / execAggregateBatch (NativeOps.h:931-942).

An aggregate carries its operands the way the reference does (``arguments`` arrays, ``indexingArguments`` ints, ``realArguments``
floats); ``Batch.getBatches`` cuts a list into launches of at most ``batchLimit``. On the GPU a batch becomes ONE launch of
native/csrc/aggregates.hip over a packed descriptor table (fp32 operands); anything else runs the same math through torch per item.
The word2vec aggregates map onto the batched skip-gram / CBOW kernel (ops/misc_kernels.py sg_cb_batch)."""
from __future__ import annotations

import struct
from typing import List, Sequence

import torch

from ..ndarray import INDArray


def _t(a):
    return a.tensor() if isinstance(a, INDArray) else a


class Aggregate:
    """Interface of the reference (Aggregate.java)."""

    def name(self) -> str:
        raise NotImplementedError

    def opNum(self) -> int:
        raise NotImplementedError


class BaseAggregate(Aggregate):
    def __init__(self):
        self.arguments: List = []
        self.shapes: List = []
        self.intArrayArguments: List = []
        self.indexingArguments: List[int] = []
        self.realArguments: List[float] = []
        self._final = None

    def getArguments(self):
        return self.arguments

    def getShapes(self):
        return self.shapes

    def getIntArrayArguments(self):
        return self.intArrayArguments

    def getIndexingArguments(self):
        return self.indexingArguments

    def getRealArguments(self):
        return self.realArguments

    def setFinalResult(self, v):
        self._final = v

    def getFinalResult(self):
        return self._final

    # launch geometry hints of the reference (kept for API parity; the HIP kernels size themselves)
    def getThreadsPerInstance(self) -> int:
        return 0

    def getSharedMemorySize(self) -> int:
        return 0

    def maxArguments(self) -> int:
        return len(self.arguments)

    def maxShapes(self) -> int:
        return 0

    def maxIntArrays(self) -> int:
        return 0

    def maxIntArraySize(self) -> int:
        return 0

    def maxIndexArguments(self) -> int:
        return len(self.indexingArguments)

    def maxRealArguments(self) -> int:
        return len(self.realArguments)

    # one item on the host / through torch
    def run(self):
        raise NotImplementedError


class AggregateAxpy(BaseAggregate):
    """y += alpha * x (opNum 2)."""

    def __init__(self, x, y, alpha: float):
        super().__init__()
        self.arguments += [x, y]
        self.indexingArguments.append(int(_t(x).numel()))
        self.realArguments.append(float(alpha))

    def name(self):
        return "aggregate_axpy"

    def opNum(self):
        return 2

    def getThreadsPerInstance(self):
        return min(768, self.indexingArguments[0])

    def run(self):
        x, y = _t(self.arguments[0]), _t(self.arguments[1])
        y.view(-1).add_(x.reshape(-1).to(y.dtype), alpha=self.realArguments[0])


class AggregateDot(BaseAggregate):
    """result = <x, y> (opNum 1); the value lands in getFinalResult()."""

    def __init__(self, x, y):
        super().__init__()
        self.arguments += [x, y]
        self.indexingArguments.append(int(_t(x).numel()))

    def name(self):
        return "aggregate_dot"

    def opNum(self):
        return 1

    def run(self):
        self.setFinalResult(float(torch.dot(_t(self.arguments[0]).reshape(-1).float(), _t(self.arguments[1]).reshape(-1).float())))


class AggregateGEMM(BaseAggregate):
    """C = alpha * op(A) . op(B) + beta * C on column-major operands with leading dimensions, cblas argument order (opNum 5).
    TransA / TransB: 111 = no transpose, 112 = transpose (CBLAS enum values, as the reference passes them); Order 102 = column major."""

    def __init__(self, Order, TransA, TransB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc):
        super().__init__()
        self.arguments += [A, B, C]
        self.indexingArguments += [int(M), int(N), int(K), int(lda), int(ldb), int(ldc), int(TransA), int(TransB), int(Order)]
        self.realArguments += [float(alpha), float(beta)]

    def name(self):
        return "aggregate_gemm"

    def opNum(self):
        return 5

    @staticmethod
    def _is_trans(v):
        return v in (112, 113, 1, ord("T"), ord("t"), True)

    def _views(self):
        M, N, K, lda, ldb, ldc, ta, tb, _ = self.indexingArguments
        ta, tb = self._is_trans(ta), self._is_trans(tb)
        A, B, C = (_t(a) for a in self.arguments)

        def cm(t, rows, cols, ld):               # column-major [rows, cols] view with leading dimension ld over the flat buffer
            return torch.as_strided(t.reshape(-1) if t.is_contiguous() else t.contiguous().view(-1), (rows, cols), (1, ld))
        a = cm(A, K if ta else M, M if ta else K, lda)
        b = cm(B, N if tb else K, K if tb else N, ldb)
        flat_c = C.view(-1) if C.is_contiguous() else None
        if flat_c is None:
            raise ValueError("AggregateGEMM: C must be a dense buffer")
        c = torch.as_strided(flat_c, (M, N), (1, ldc))
        return (a.t() if ta else a), (b.t() if tb else b), c

    def run(self):
        a, b, c = self._views()
        alpha, beta = self.realArguments
        r = alpha * (a.float() @ b.float())
        c.copy_((r + beta * c.float()).to(c.dtype) if beta != 0.0 else r.to(c.dtype))


class _W2VAggregate(BaseAggregate):
    """Shared shape of the word2vec aggregates: tables syn0 / syn1 / syn1Neg (+ expTable, negTable kept for signature parity), one
    training item (w1 = the word whose vector moves, Huffman points / codes, negative-sampling target and rounds)."""
    CBOW = False

    def __init__(self, syn0, syn1, syn1Neg, expTable, negTable, idxSyn0, idxSyn1: Sequence[int], codes: Sequence[int], negativeRounds,
                 ngStarter, vectorLength, alpha, nextRandom, vocabSize, context: Sequence[int] = (), inferenceVector=None):
        super().__init__()
        self.arguments += [syn0, syn1, expTable, syn1Neg, negTable, inferenceVector]
        self.intArrayArguments += [list(idxSyn1), list(codes)] + ([list(context)] if self.CBOW else [])
        self.indexingArguments += [int(idxSyn0), int(vectorLength), len(idxSyn1), int(negativeRounds), 0 if negTable is None else int(_t(negTable).numel()),
                                   int(vocabSize), int(ngStarter)]
        self.realArguments += [float(alpha), float(int(nextRandom))]
        self.nextRandom = int(nextRandom)


class AggregateSkipGram(_W2VAggregate):
    def name(self):
        return "aggregate_skipgram"

    def opNum(self):
        return 3


class AggregateCBOW(_W2VAggregate):
    CBOW = True

    def name(self):
        return "aggregate_cbow"

    def opNum(self):
        return 4


class HierarchicSoftmax(_W2VAggregate):
    """Hierarchic-softmax-only round (opNum 0): a skip-gram item without negative sampling."""

    def __init__(self, syn0, syn1, expTable, idxSyn0, idxSyn1, codes, vectorLength, alpha, vocabSize):
        super().__init__(syn0, syn1, None, expTable, None, idxSyn0, idxSyn1, codes, 0, 0, vectorLength, alpha, 0, vocabSize)

    def name(self):
        return "aggregate_hs"

    def opNum(self):
        return 0


class Batch:
    """Aggregates of ONE op executed together (Batch.java); ``getBatches`` partitions a list into launches of at most ``batchLimit``."""
    batchLimit = 512

    def __init__(self, aggregates: List[Aggregate]):
        if not aggregates:
            raise ValueError("Batch needs at least one aggregate")
        if len({a.opNum() for a in aggregates}) != 1:
            raise ValueError("all aggregates of a batch must be the same op")
        self.aggregates = list(aggregates)
        self.sample = aggregates[0]

    @classmethod
    def getBatchLimit(cls):
        return cls.batchLimit

    def getAggregates(self):
        return self.aggregates

    def getSample(self):
        return self.sample

    def getNumAggregates(self):
        return len(self.aggregates)

    def opNum(self):
        return self.sample.opNum()

    def isFull(self):
        return len(self.aggregates) >= self.batchLimit

    def append(self, aggregate) -> bool:
        if self.isFull() or aggregate.opNum() != self.opNum():
            return False
        self.aggregates.append(aggregate)
        return True

    @classmethod
    def getBatches(cls, aggregates: List[Aggregate], partitionSize: int = None) -> List["Batch"]:
        n = partitionSize or cls.batchLimit
        return [Batch(aggregates[i:i + n]) for i in range(0, len(aggregates), n)]


# ------------------------------------------------------------------------------------------------------------------ execution
STATS = {"launches": 0, "items_native": 0, "items_generic": 0}


def _native_blas_ok(items) -> bool:
    from ...native import lib as L
    ts = [_t(a) for it in items for a in it.arguments]
    return bool(ts) and all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts) \
        and L.use_hip(ts[0]) and len({t.device for t in ts}) == 1


def _launch(op, table: bytes, count, dev, out=None, max_tiles=1):
    from ...native import lib as L
    buf = torch.frombuffer(bytearray(table), dtype=torch.uint8).to(dev, non_blocking=False)
    L.check(L.hip().dl4j_aggregate_batch(L.c_int(op), L.ptr(buf), L.ptr(out), L.c_int(count), L.c_int(max_tiles), L.stream()), "aggregate_batch")
    STATS["launches"] += 1
    STATS["items_native"] += count
    return buf


def _exec_w2v(items: List[_W2VAggregate]):
    """Skip-gram / hierarchic-softmax batches become ONE `skipgram` custom-op call over the whole batch (batched word2vec kernel on the
    GPU, the generic loop on the host); CBOW items go through the `cbow` op."""
    from .custom import exec_op
    s = items[0]
    syn0, syn1, expT, syn1n, negT, _ = (None if a is None else _t(a) for a in s.arguments)
    dev = syn0.device
    empty = torch.empty(0, device=dev, dtype=syn0.dtype)
    syn1 = empty if syn1 is None else syn1
    syn1n = empty if syn1n is None else syn1n
    negT = torch.empty(0, device=dev, dtype=torch.int32) if negT is None else negT
    expT = empty if expT is None else expT
    rounds = s.indexingArguments[3]
    if not s.CBOW and all(it.arguments[0] is s.arguments[0] for it in items):
        B = len(items)
        code_len = max(max(len(it.intArrayArguments[0]) for it in items), 1)
        idx = torch.full((B, code_len), -1, dtype=torch.int32)
        codes = torch.zeros((B, code_len), dtype=torch.int8)
        for r, it in enumerate(items):
            n = len(it.intArrayArguments[0])
            if n:
                idx[r, :n] = torch.tensor(it.intArrayArguments[0], dtype=torch.int32)
                codes[r, :n] = torch.tensor(it.intArrayArguments[1], dtype=torch.int8)
        target = torch.tensor([it.indexingArguments[0] for it in items], dtype=torch.int32, device=dev)
        ngs = torch.tensor([it.indexingArguments[6] for it in items], dtype=torch.int32, device=dev)
        alpha = torch.tensor([it.realArguments[0] for it in items], dtype=torch.float32, device=dev)
        rnd = torch.tensor([it.nextRandom for it in items], dtype=torch.int64, device=dev)
        exec_op("skipgram", [target, ngs, idx.to(dev), codes.to(dev), syn0, syn1, syn1n, expT, negT, alpha, rnd, empty], iArgs=[1, rounds],
                bArgs=[False, False])
        STATS["launches"] += 1
        STATS["items_native" if syn0.is_cuda else "items_generic"] += B
        return
    for it in items:
        t0 = _t(it.arguments[0])
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=t0.device)            # noqa: E731
        ins = [i32(it.indexingArguments[0]), i32(it.indexingArguments[6])]
        if it.CBOW:
            ins.append(i32(it.intArrayArguments[2]))
        ins += [i32(it.intArrayArguments[0]), torch.tensor(it.intArrayArguments[1], dtype=torch.int8, device=t0.device), t0, syn1, syn1n, expT, negT,
                torch.tensor(it.realArguments[0], device=t0.device), torch.tensor(it.nextRandom, dtype=torch.int64, device=t0.device)]
        if it.CBOW:
            ins += [torch.tensor(0, dtype=torch.int32, device=t0.device), torch.empty(0, dtype=torch.int32, device=t0.device)]
        ins.append(empty)                                                                # inference vector slot (unused: training)
        exec_op("cbow" if it.CBOW else "skipgram", ins, iArgs=[1, rounds], bArgs=[True, False] if it.CBOW else [False, False])
        STATS["items_generic"] += 1


def exec_batch(batch: Batch):
    """OpExecutioner.exec(Batch): the whole batch in one launch where the operands allow it."""
    items = batch.getAggregates()
    op = batch.opNum()
    if op in (0, 3, 4):
        return _exec_w2v(items)
    if op in (1, 2, 5) and _native_blas_ok(items):
        dev = _t(items[0].arguments[0]).device
        if op == 2:
            tab = b"".join(struct.pack("<QQfi", _t(it.arguments[0]).data_ptr(), _t(it.arguments[1]).data_ptr(), it.realArguments[0],
                                       it.indexingArguments[0]) for it in items)
            _launch(2, tab, len(items), dev)
            return
        if op == 1:
            tab = b"".join(struct.pack("<QQii", _t(it.arguments[0]).data_ptr(), _t(it.arguments[1]).data_ptr(), it.indexingArguments[0], 0)
                           for it in items)
            out = torch.empty(len(items), device=dev, dtype=torch.float32)
            _launch(1, tab, len(items), dev, out)
            for it, v in zip(items, out.tolist()):
                it.setFinalResult(v)
            return
        rows, tiles = [], 1
        for it in items:
            M, N, K, lda, ldb, ldc, ta, tb, order = it.indexingArguments
            if order not in (102, ord("f"), ord("F"), 0):
                break
            ta, tb = AggregateGEMM._is_trans(ta), AggregateGEMM._is_trans(tb)
            A, B, C = (_t(a) for a in it.arguments)
            need_a = ((M if ta else K) - 1) * lda + (K if ta else M)
            need_b = ((K if tb else N) - 1) * ldb + (N if tb else K)
            if A.numel() < need_a or B.numel() < need_b or C.numel() < (N - 1) * ldc + M:
                raise ValueError("AggregateGEMM: operand buffer smaller than its (rows, columns, leading dimension) describe")
            rows.append(struct.pack("<QQQ8i2f", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, lda, ldb, ldc, int(ta), int(tb),
                                    it.realArguments[0], it.realArguments[1]))
            tiles = max(tiles, ((M + 15) // 16) * ((N + 15) // 16))
        else:
            _launch(5, b"".join(rows), len(items), dev, max_tiles=tiles)
            return
    for it in items:
        it.run()
        STATS["items_generic"] += 1


def exec_aggregate(aggregate: Aggregate):
    """OpExecutioner.exec(Aggregate): a batch of one."""
    exec_batch(Batch([aggregate]))
    return aggregate.getFinalResult()
