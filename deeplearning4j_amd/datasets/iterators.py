"""DataSetIterators (reference: nd4j-api/.../linalg/dataset/api/iterator/*, AsyncDataSetIterator.java:366-423,
deeplearning4j-utility-iterators/.../BenchmarkDataSetIterator.java and friends).

``AsyncDataSetIterator`` = background prefetch thread + bounded queue; on MI355X the thread also
stages batches into pinned host memory and issues the H2D copy on a dedicated copy stream so the
transfer overlaps compute (the reference pins the prefetch thread to the consumer's device)."""
from __future__ import annotations

import queue
import threading
from typing import List, Optional

import torch

from .dataset import DataSet, MultiDataSet


class DataSetIterator:
    preProcessor = None

    def hasNext(self):
        raise NotImplementedError

    def next(self, num=None):
        raise NotImplementedError

    def reset(self):
        pass

    def resetSupported(self):
        return True

    def asyncSupported(self):
        return True

    def batch(self):
        return -1

    def inputColumns(self):
        return -1

    def totalOutcomes(self):
        return -1

    def getLabels(self):
        return None

    def setPreProcessor(self, p):
        self.preProcessor = p

    def getPreProcessor(self):
        return self.preProcessor

    def _pp(self, ds):
        if self.preProcessor is not None:
            self.preProcessor.preProcess(ds)
        return ds

    def __iter__(self):
        self.reset()
        while self.hasNext():
            yield self.next()


MultiDataSetIterator = DataSetIterator


class ShardedDataSetIterator(DataSetIterator):
    """This rank's share of a base iterator for data-parallel training: batch k of every COMPLETE round of ``world``
    consecutive batches goes to rank ``k % world``; a ragged tail (fewer than ``world`` batches left) is dropped on every
    rank so all ranks run the same number of iterations and every collective has all its peers.

    Iterators that can address batches (``numBatches()`` + ``skip(n)``, e.g. list / benchmark iterators) never materialise
    a foreign batch; for plain streams the foreign batches of a round are pulled and discarded before preprocessing."""

    def __init__(self, base, rank, world):
        self.base, self.rank, self.world = base, int(rank), max(1, int(world))
        self._held = None

    def _fill(self):
        if self._held is not None or self.world == 1:
            return
        b = self.base
        if hasattr(b, "numBatches") and hasattr(b, "skip"):
            left = b.numBatches() - b.position()
            if left < self.world:
                b.skip(left)
                return
            b.skip(self.rank)
            self._held = b.next()
            b.skip(self.world - self.rank - 1)
            return
        mine, got = None, 0
        while got < self.world and b.hasNext():
            d = b.next()
            if got == self.rank:
                mine = d
            got += 1
        if got == self.world:
            self._held = mine

    def hasNext(self):
        if self.world == 1:
            return self.base.hasNext()
        self._fill()
        return self._held is not None

    def next(self, num=None):
        if self.world == 1:
            return self.base.next()
        self._fill()
        d, self._held = self._held, None
        if d is None:
            raise StopIteration
        return d

    def reset(self):
        self._held = None
        self.base.reset()

    def resetSupported(self):
        return self.base.resetSupported()

    def asyncSupported(self):
        return False

    def batch(self):
        return self.base.batch()


class ListDataSetIterator(DataSetIterator):
    def __init__(self, data, batchSize=None):
        if isinstance(data, DataSet):
            data = data.batchBy(batchSize or data.numExamples())
        elif batchSize is not None and data and data[0].numExamples() == 1:
            merged = DataSet.merge(list(data))
            data = merged.batchBy(batchSize)
        self._data, self._i, self._bs = list(data), 0, batchSize or (data[0].numExamples() if data else 0)

    def hasNext(self):
        return self._i < len(self._data)

    def next(self, num=None):
        d = self._data[self._i]
        self._i += 1
        return self._pp(d)

    def reset(self):
        self._i = 0

    # random access (lets ShardedDataSetIterator skip foreign batches without touching them)
    def numBatches(self):
        return len(self._data)

    def position(self):
        return self._i

    def skip(self, n):
        self._i = min(len(self._data), self._i + int(n))

    def batch(self):
        return self._bs

    def inputColumns(self):
        return self._data[0].numInputs()

    def totalOutcomes(self):
        return self._data[0].numOutcomes()


class INDArrayDataSetIterator(ListDataSetIterator):
    def __init__(self, features, labels, batchSize):
        super().__init__(DataSet(features, labels), batchSize)


class ExistingDataSetIterator(DataSetIterator):
    def __init__(self, iterable):
        self._src = iterable
        self._it = iter(iterable)
        self._nxt = None

    def hasNext(self):
        if self._nxt is None:
            try:
                self._nxt = next(self._it)
            except StopIteration:
                return False
        return True

    def next(self, num=None):
        self.hasNext()
        d, self._nxt = self._nxt, None
        return self._pp(d)

    def reset(self):
        self._it = iter(self._src)
        self._nxt = None


class BenchmarkDataSetIterator(DataSetIterator):
    """A single synthetic batch replayed ``totalIterations`` times (BenchmarkDataSetIterator.java)."""

    def __init__(self, featuresShape, numLabels, totalIterations, device=None, dtype=torch.float32, seed=42,
                 labelsShape=None):
        g = torch.Generator().manual_seed(seed)
        f = torch.rand(tuple(featuresShape), generator=g, dtype=torch.float32).to(dtype)
        mb = featuresShape[0]
        if labelsShape is None:
            idx = torch.randint(0, numLabels, (mb,), generator=g)
            l = torch.zeros(mb, numLabels, dtype=dtype)
            l[torch.arange(mb), idx] = 1.0
        else:
            l = torch.rand(tuple(labelsShape), generator=g).to(dtype)
        self._ds = DataSet(f, l)
        if device is not None:
            self._ds = self._ds.to(device, non_blocking=False)
        self._limit, self._n = int(totalIterations), 0

    def hasNext(self):
        return self._n < self._limit

    def next(self, num=None):
        self._n += 1
        return self._ds

    def reset(self):
        self._n = 0

    def batch(self):
        return self._ds.numExamples()

    def asyncSupported(self):
        return False


class BenchmarkMultiDataSetIterator(BenchmarkDataSetIterator):
    def next(self, num=None):
        self._n += 1
        return MultiDataSet([self._ds.features], [self._ds.labels])


class EarlyTerminationDataSetIterator(DataSetIterator):
    def __init__(self, underlying, terminationPoint):
        self._u, self._n, self._limit = underlying, 0, int(terminationPoint)

    def hasNext(self):
        return self._n < self._limit and self._u.hasNext()

    def next(self, num=None):
        self._n += 1
        return self._u.next()

    def reset(self):
        self._n = 0
        self._u.reset()


EarlyTerminationMultiDataSetIterator = EarlyTerminationDataSetIterator


class MultipleEpochsIterator(DataSetIterator):
    def __init__(self, numEpochs, underlying):
        self._u, self._epochs, self._e = underlying, int(numEpochs), 0

    def hasNext(self):
        if self._u.hasNext():
            return True
        if self._e + 1 < self._epochs:
            self._e += 1
            self._u.reset()
            return self._u.hasNext()
        return False

    def next(self, num=None):
        return self._u.next()

    def reset(self):
        self._e = 0
        self._u.reset()


class SamplingDataSetIterator(DataSetIterator):
    def __init__(self, dataset, batchSize, totalNumberSamples, seed=None):
        self._ds, self._bs, self._tot, self._n, self._seed = dataset, batchSize, totalNumberSamples, 0, seed

    def hasNext(self):
        return self._n < self._tot

    def next(self, num=None):
        self._n += self._bs
        return self._pp(self._ds.sample(self._bs, withReplacement=True))

    def reset(self):
        self._n = 0


class IteratorDataSetIterator(DataSetIterator):
    """Re-batches a stream of DataSets of arbitrary size into fixed-size minibatches."""

    def __init__(self, iterator, batchSize):
        self._src, self._bs = iterator, batchSize
        self._buf: List[DataSet] = []
        self._count = 0

    def _fill(self):
        while self._count < self._bs and self._src.hasNext():
            d = self._src.next()
            self._buf.append(d)
            self._count += d.numExamples()

    def hasNext(self):
        self._fill()
        return self._count > 0

    def next(self, num=None):
        self._fill()
        m = DataSet.merge(self._buf)
        out = m.getRange(0, min(self._bs, m.numExamples()))
        rest = m.getRange(self._bs, m.numExamples()) if m.numExamples() > self._bs else None
        self._buf = [rest] if rest is not None else []
        self._count = rest.numExamples() if rest is not None else 0
        return self._pp(out)

    def reset(self):
        self._src.reset()
        self._buf, self._count = [], 0


class JointParallelDataSetIterator(DataSetIterator):
    """Round-robins over several source iterators (JointParallelDataSetIterator.java)."""

    def __init__(self, *iterators, stopIfFirstFinished=False):
        self._its, self._k, self._stop = list(iterators), 0, stopIfFirstFinished

    def hasNext(self):
        if self._stop:
            return all(i.hasNext() for i in self._its)
        return any(i.hasNext() for i in self._its)

    def next(self, num=None):
        for _ in range(len(self._its)):
            it = self._its[self._k % len(self._its)]
            self._k += 1
            if it.hasNext():
                return it.next()
        raise StopIteration

    def reset(self):
        self._k = 0
        for i in self._its:
            i.reset()


class AsyncDataSetIterator(DataSetIterator):
    """Background prefetch (queue >= 2) with pinned staging + H2D on a copy stream."""
    _END = object()

    def __init__(self, underlying, queueSize=2, device=None, pin=True):
        self._u = underlying
        self._qsize = max(2, int(queueSize))
        self._device = torch.device(device) if device is not None else None
        self._pin = pin and self._device is not None and self._device.type == "cuda"
        self._copy_stream = torch.cuda.Stream(self._device) if self._pin else None
        self._q = None
        self._thread = None
        self._peek = None
        self._err = None
        self._stop = threading.Event()
        self._start()

    def _start(self):
        self._stop.clear()
        self._q = queue.Queue(self._qsize)
        self._peek = None
        self._thread = threading.Thread(target=self._run, daemon=True, name="ADSI prefetch thread")
        self._thread.start()

    def _run(self):
        try:
            if self._device is not None and self._device.type == "cuda":
                torch.cuda.set_device(self._device)
            while not self._stop.is_set() and self._u.hasNext():
                ds = self._u.next()
                if self._device is not None:
                    if self._pin:
                        ds = ds.pin_memory()
                        with torch.cuda.stream(self._copy_stream):
                            ds = ds.to(self._device, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(self._copy_stream)
                        ds._ready = ev
                    else:
                        ds = ds.to(self._device)
                while not self._stop.is_set():
                    try:
                        self._q.put(ds, timeout=0.1)
                        break
                    except queue.Full:
                        continue
        except BaseException as e:  # surface errors in the consumer thread
            self._err = e
        finally:
            while True:
                try:
                    self._q.put(self._END, timeout=0.1)
                    break
                except queue.Full:
                    if self._stop.is_set():
                        break

    def hasNext(self):
        if self._peek is None:
            self._peek = self._q.get()
        if self._peek is self._END:
            if self._err is not None:
                e, self._err = self._err, None
                raise e
            return False
        return True

    def next(self, num=None):
        if not self.hasNext():
            raise StopIteration
        ds, self._peek = self._peek, None
        ev = getattr(ds, "_ready", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        return self._pp(ds)

    def reset(self):
        self.shutdown()
        self._u.reset()
        self._start()

    def shutdown(self):
        self._stop.set()
        if self._thread is not None:
            while self._thread.is_alive():
                try:
                    self._q.get_nowait()
                except queue.Empty:
                    pass
                self._thread.join(timeout=0.05)
        self._thread = None

    def asyncSupported(self):
        return False

    def batch(self):
        return self._u.batch()


AsyncMultiDataSetIterator = AsyncDataSetIterator


class IrisDataSetIterator(ListDataSetIterator):
    """Synthetic Iris-shaped data (150 x 4 -> 3 classes); the real CSV is a downloaded resource in the
    reference (IrisDataSetIterator.java) which is unavailable offline — class-separable Gaussians instead."""

    def __init__(self, batch=150, numExamples=150, seed=6):
        g = torch.Generator().manual_seed(seed)
        means = torch.tensor([[5.0, 3.4, 1.5, 0.2], [5.9, 2.8, 4.3, 1.3], [6.6, 3.0, 5.6, 2.0]])
        per = numExamples // 3
        xs, ys = [], []
        for c in range(3):
            xs.append(means[c] + 0.3 * torch.randn(per, 4, generator=g))
            y = torch.zeros(per, 3)
            y[:, c] = 1.0
            ys.append(y)
        ds = DataSet(torch.cat(xs), torch.cat(ys))
        ds.shuffle(seed)
        super().__init__(ds, batch)
