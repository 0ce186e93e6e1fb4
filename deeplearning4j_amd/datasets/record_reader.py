"""DataVec -> DataSet bridges (reference: deeplearning4j-data/deeplearning4j-datavec-iterators/.../datasets/datavec/
{RecordReaderDataSetIterator,SequenceRecordReaderDataSetIterator,RecordReaderMultiDataSetIterator}.java)."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..datavec.records import RecordReader, SequenceRecordReader
from ..datavec.writable import NDArrayWritable, Writable
from .dataset import DataSet, MultiDataSet
from .iterators import DataSetIterator


def _to_float(w: Writable) -> float:
    return w.toDouble()


class RecordReaderDataSetIterator(DataSetIterator):
    """Classification: ``labelIndex`` + ``numPossibleLabels`` (one-hot); regression: ``labelIndexFrom..labelIndexTo`` with
    ``regression=True``; ``labelIndex=-1``: features only.  NDArrayWritable records (images) pass through as tensors."""

    def __init__(self, recordReader: RecordReader, batchSize, labelIndex=-1, numPossibleLabels=-1, labelIndexTo=None, regression=False,
                 maxNumBatches=-1, collectMetaData=False):
        self.rr, self.batchSize = recordReader, int(batchSize)
        self.labelFrom, self.labelTo = labelIndex, (labelIndex if labelIndexTo is None else labelIndexTo)
        self.numLabels, self.regression, self.maxBatches = numPossibleLabels, regression, maxNumBatches
        self.collectMetaData = collectMetaData
        self._count = 0
        if not regression and labelIndex >= 0 and numPossibleLabels < 1:
            labs = recordReader.getLabels() if hasattr(recordReader, "getLabels") else None
            if labs:
                self.numLabels = len(labs)
            else:
                raise ValueError("numPossibleLabels must be set for classification (labelIndex=%d)" % labelIndex)

    class Builder:
        def __init__(self, rr, batchSize):
            self._rr, self._bs, self._kw = rr, batchSize, {}

        def classification(self, labelIndex, numClasses):
            self._kw.update(labelIndex=labelIndex, numPossibleLabels=numClasses)
            return self

        def regression(self, frm, to=None):
            self._kw.update(labelIndex=frm, labelIndexTo=frm if to is None else to, regression=True)
            return self

        def maxNumBatches(self, n):
            self._kw["maxNumBatches"] = n
            return self

        def collectMetaData(self, b):
            self._kw["collectMetaData"] = b
            return self

        def preProcessor(self, p):
            self._pre = p
            return self

        def build(self):
            it = RecordReaderDataSetIterator(self._rr, self._bs, **self._kw)
            if hasattr(self, "_pre"):
                it.setPreProcessor(self._pre)
            return it

    def hasNext(self):
        return self.rr.hasNext() and (self.maxBatches < 0 or self._count < self.maxBatches)

    def reset(self):
        self.rr.reset()
        self._count = 0

    def batch(self):
        return self.batchSize

    def totalOutcomes(self):
        return self.numLabels if not self.regression else (self.labelTo - self.labelFrom + 1)

    def getLabels(self):
        return self.rr.getLabels() if hasattr(self.rr, "getLabels") else None

    def _split_record(self, rec: List[Writable]):
        if len(rec) >= 1 and isinstance(rec[0], NDArrayWritable):
            x = torch.as_tensor(rec[0].value).float()
            x = x if x.dim() > 1 else x.unsqueeze(0)
            y = None
            if len(rec) > 1:
                if isinstance(rec[1], NDArrayWritable):
                    y = torch.as_tensor(rec[1].value).float()
                elif self.regression:
                    y = torch.tensor([[_to_float(w) for w in rec[1:]]], dtype=torch.float32)
                else:
                    y = torch.zeros(1, self.numLabels)
                    y[0, rec[1].toInt()] = 1
            return x, y
        vals = np.array([_to_float(w) for w in rec], dtype=np.float32)
        if self.labelFrom < 0:
            return torch.from_numpy(vals).unsqueeze(0), None
        lf, lt = self.labelFrom, self.labelTo
        feat = np.concatenate([vals[:lf], vals[lt + 1:]])
        if self.regression:
            y = torch.from_numpy(vals[lf:lt + 1].copy()).unsqueeze(0)
        else:
            k = int(vals[lf])
            if not 0 <= k < self.numLabels:
                raise ValueError("Invalid classification data: expect label value (at label index column = %d) to be in range 0 to %d "
                                 "inclusive (0 to numClasses-1, with numClasses=%d); got label value of %d" % (lf, self.numLabels - 1, self.numLabels, k))
            y = torch.zeros(1, self.numLabels)
            y[0, k] = 1
        return torch.from_numpy(feat).unsqueeze(0), y

    def next(self, num=None):
        n = num or self.batchSize
        xs, ys, metas = [], [], []
        while len(xs) < n and self.rr.hasNext():
            if self.collectMetaData:
                r = self.rr.nextRecord()
                rec, meta = r.getRecord(), r.getMetaData()
                metas.append(meta)
            else:
                rec = self.rr.next()
            x, y = self._split_record(rec)
            xs.append(x)
            if y is not None:
                ys.append(y)
        self._count += 1
        ds = DataSet(torch.cat(xs, 0), torch.cat(ys, 0) if ys else None)
        if self.collectMetaData:
            ds.exampleMetaData = metas
        return self._pp(ds)

    def loadFromMetaData(self, metas):
        want = {(m.uri, m.index) for m in metas}
        self.rr.reset()
        xs, ys = [], []
        while self.rr.hasNext():
            r = self.rr.nextRecord()
            m = r.getMetaData()
            if m is not None and (m.uri, m.index) in want:
                x, y = self._split_record(r.getRecord())
                xs.append(x)
                if y is not None:
                    ys.append(y)
        return DataSet(torch.cat(xs, 0), torch.cat(ys, 0) if ys else None)


class SequenceRecordReaderDataSetIterator(DataSetIterator):
    """Time-series [mb, size, T] with masks.  Either one reader (label column inside each step) or separate feature / label
    readers; ``alignmentMode``: EQUAL_LENGTH, ALIGN_START, ALIGN_END (SequenceRecordReaderDataSetIterator.java)."""

    class AlignmentMode:
        EQUAL_LENGTH, ALIGN_START, ALIGN_END = "EQUAL_LENGTH", "ALIGN_START", "ALIGN_END"

    def __init__(self, featuresReader: SequenceRecordReader, labelsReader_or_batch, miniBatchSize=None, numPossibleLabels=-1, labelIndex=None,
                 regression=False, alignmentMode="EQUAL_LENGTH"):
        if isinstance(labelsReader_or_batch, SequenceRecordReader):
            self.fr, self.lr, self.batchSize = featuresReader, labelsReader_or_batch, int(miniBatchSize)
            self.labelIndex = None
        else:
            # single-reader form: (reader, miniBatchSize, numPossibleLabels, labelIndex, regression)
            self.fr, self.lr, self.batchSize = featuresReader, None, int(labelsReader_or_batch)
            if miniBatchSize is not None and labelIndex is None:
                numPossibleLabels, labelIndex = miniBatchSize, numPossibleLabels
            self.labelIndex = labelIndex
        self.numLabels, self.regression, self.alignment = numPossibleLabels, regression, alignmentMode

    def hasNext(self):
        return self.fr.hasNext()

    def reset(self):
        self.fr.reset()
        if self.lr is not None:
            self.lr.reset()

    def batch(self):
        return self.batchSize

    def totalOutcomes(self):
        return self.numLabels

    def _label_steps(self, steps):
        if self.regression:
            return np.array([[w.toDouble() for w in s] for s in steps], dtype=np.float32)
        out = np.zeros((len(steps), self.numLabels), dtype=np.float32)
        for t, s in enumerate(steps):
            out[t, s[0].toInt()] = 1
        return out

    def next(self, num=None):
        n = num or self.batchSize
        feats, labs = [], []
        while len(feats) < n and self.fr.hasNext():
            fseq = self.fr.sequenceRecord()
            if self.lr is not None:
                lseq = self.lr.sequenceRecord()
                f = np.array([[w.toDouble() for w in s] for s in fseq], dtype=np.float32)
                l = self._label_steps(lseq)
            else:
                li = self.labelIndex
                f = np.array([[w.toDouble() for k, w in enumerate(s) if k != li] for s in fseq], dtype=np.float32)
                l = self._label_steps([[s[li]] for s in fseq])
            feats.append(f)
            labs.append(l)
        T = max(max(f.shape[0] for f in feats), max(l.shape[0] for l in labs))
        mb = len(feats)
        X = torch.zeros(mb, feats[0].shape[1], T)
        Y = torch.zeros(mb, labs[0].shape[1], T)
        fm, lm = torch.zeros(mb, T), torch.zeros(mb, T)
        for i, (f, l) in enumerate(zip(feats, labs)):
            if self.alignment == self.AlignmentMode.EQUAL_LENGTH and f.shape[0] != l.shape[0]:
                raise ValueError("Alignment mode is EQUAL_LENGTH but sequence %d has %d feature steps and %d label steps" % (i, f.shape[0], l.shape[0]))
            fs = T - f.shape[0] if self.alignment == self.AlignmentMode.ALIGN_END else 0
            ls = T - l.shape[0] if self.alignment == self.AlignmentMode.ALIGN_END else 0
            X[i, :, fs:fs + f.shape[0]] = torch.from_numpy(f).t()
            Y[i, :, ls:ls + l.shape[0]] = torch.from_numpy(l).t()
            fm[i, fs:fs + f.shape[0]] = 1
            lm[i, ls:ls + l.shape[0]] = 1
        need_masks = bool((fm == 0).any() or (lm == 0).any())
        return self._pp(DataSet(X, Y, fm if need_masks else None, lm if need_masks else None))


class RecordReaderMultiDataSetIterator(DataSetIterator):
    """Multiple named readers -> multiple input / output arrays by column ranges (RecordReaderMultiDataSetIterator.java)."""

    def __init__(self, batchSize, readers: Dict[str, RecordReader], seqReaders: Dict[str, SequenceRecordReader], inputs, outputs):
        self.batchSize, self.readers, self.seqReaders, self.inputs, self.outputs = batchSize, readers, seqReaders, inputs, outputs

    class Builder:
        def __init__(self, batchSize):
            self._bs, self._r, self._s, self._in, self._out = batchSize, {}, {}, [], []

        def addReader(self, name, rr):
            self._r[name] = rr
            return self

        def addSequenceReader(self, name, rr):
            self._s[name] = rr
            return self

        def addInput(self, name, frm=None, to=None):
            self._in.append(("range", name, frm, to))
            return self

        def addInputOneHot(self, name, column, numClasses):
            self._in.append(("onehot", name, column, numClasses))
            return self

        def addOutput(self, name, frm=None, to=None):
            self._out.append(("range", name, frm, to))
            return self

        def addOutputOneHot(self, name, column, numClasses):
            self._out.append(("onehot", name, column, numClasses))
            return self

        def build(self):
            return RecordReaderMultiDataSetIterator(self._bs, self._r, self._s, self._in, self._out)

    def hasNext(self):
        return all(r.hasNext() for r in list(self.readers.values()) + list(self.seqReaders.values()))

    def reset(self):
        for r in list(self.readers.values()) + list(self.seqReaders.values()):
            r.reset()

    def batch(self):
        return self.batchSize

    @staticmethod
    def _pick(kind, rec, a, b):
        if kind == "onehot":
            v = np.zeros(b, dtype=np.float32)
            v[rec[a].toInt()] = 1
            return v
        lo = 0 if a is None else a
        hi = len(rec) - 1 if b is None else b
        return np.array([w.toDouble() for w in rec[lo:hi + 1]], dtype=np.float32)

    def next(self, num=None):
        n = num or self.batchSize
        rows: List[Dict[str, object]] = []
        while len(rows) < n and self.hasNext():
            row = {k: r.next() for k, r in self.readers.items()}
            row.update({k: r.sequenceRecord() for k, r in self.seqReaders.items()})
            rows.append(row)

        def build(specs):
            arrs, masks = [], []
            for kind, name, a, b in specs:
                if name in self.readers:
                    arrs.append(torch.from_numpy(np.stack([self._pick(kind, row[name], a, b) for row in rows])))
                    masks.append(None)
                else:
                    seqs = [np.stack([self._pick(kind, step, a, b) for step in row[name]]) for row in rows]
                    T = max(s.shape[0] for s in seqs)
                    X = torch.zeros(len(seqs), seqs[0].shape[1], T)
                    M = torch.zeros(len(seqs), T)
                    for i, s in enumerate(seqs):
                        X[i, :, :s.shape[0]] = torch.from_numpy(s).t()
                        M[i, :s.shape[0]] = 1
                    arrs.append(X)
                    masks.append(M if bool((M == 0).any()) else None)
            return arrs, (masks if any(m is not None for m in masks) else None)
        fx, fm = build(self.inputs)
        ly, lm = build(self.outputs)
        return self._pp(MultiDataSet(fx, ly, fm, lm))
