"""Input splits + record readers (reference: datavec-api/.../split/*.java, records/reader/impl/**:
csv/{CSVRecordReader,CSVSequenceRecordReader,CSVRegexRecordReader}, LineRecordReader, misc/{LibSvmRecordReader,SVMLightRecordReader},
regex/RegexLineRecordReader, jackson/JacksonLineRecordReader, collection/{CollectionRecordReader,CollectionSequenceRecordReader},
FileRecordReader, inmemory/InMemoryRecordReader, transform/TransformProcessRecordReader).

All-numeric CSV / SVMLight files have a native bulk path (``native/csrc/rt_io.cpp`` dv_csv_parse / dv_svmlight_parse): one C++
pass over the file straight into a float32 matrix, used by ``loadAllNumeric`` and by RecordReaderDataSetIterator."""
from __future__ import annotations

import csv
import glob
import io
import json
import os
import random
import re
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch

from .writable import DoubleWritable, IntWritable, Text, Writable, wrap


# ----------------------------------------------------------------------------- input splits
class InputSplit:
    def locations(self) -> List[str]:
        raise NotImplementedError

    def length(self):
        return len(self.locations())

    def reset(self):
        pass

    def sample(self, weights: Sequence[float], rng_seed=None):
        """Split into len(weights) InputSplits (BaseInputSplit.sample)."""
        locs = list(self.locations())
        random.Random(rng_seed).shuffle(locs)
        tot = float(sum(weights))
        out, start = [], 0
        for i, w in enumerate(weights):
            n = int(round(len(locs) * w / tot)) if i < len(weights) - 1 else len(locs) - start
            out.append(CollectionInputSplit(locs[start:start + n]))
            start += n
        return out


class FileSplit(InputSplit):
    def __init__(self, root, allowFormats: Optional[Sequence[str]] = None, recursive=True, rng_seed=None):
        self.root, self.allow, self.recursive, self.seed = os.fspath(root), allowFormats, recursive, rng_seed
        self._locs = None

    def locations(self):
        if self._locs is None:
            if os.path.isfile(self.root):
                locs = [self.root]
            elif self.recursive:
                locs = [os.path.join(d, f) for d, _, fs in os.walk(self.root) for f in fs]
            else:
                locs = [os.path.join(self.root, f) for f in os.listdir(self.root) if os.path.isfile(os.path.join(self.root, f))]
            if self.allow:
                al = tuple("." + a.lower().lstrip(".") for a in self.allow)
                locs = [l for l in locs if l.lower().endswith(al)]
            locs.sort()
            if self.seed is not None:
                random.Random(self.seed).shuffle(locs)
            self._locs = locs
        return self._locs

    def getRootDir(self):
        return self.root


class CollectionInputSplit(InputSplit):
    def __init__(self, uris: Iterable[str]):
        self._locs = [os.fspath(u) for u in uris]

    def locations(self):
        return self._locs


class NumberedFileInputSplit(InputSplit):
    """``baseString`` with one ``%d`` placeholder, indices minIdx..maxIdx inclusive."""

    def __init__(self, baseString, minIdx, maxIdx):
        if "%d" not in baseString and not re.search(r"%0\d+d", baseString):
            raise ValueError("Base String must match this regular expression: .*%(0\\d)?d.*")
        self.base, self.min, self.max = baseString, int(minIdx), int(maxIdx)

    def locations(self):
        return [self.base % i for i in range(self.min, self.max + 1)]


class StringSplit(InputSplit):
    def __init__(self, data: str):
        self.data = data

    def locations(self):
        return []


class ListStringSplit(InputSplit):
    def __init__(self, data: List[List[str]]):
        self.data = data

    def locations(self):
        return []


# ----------------------------------------------------------------------------- base readers
class RecordMetaData:
    def __init__(self, uri, index):
        self.uri, self.index = uri, index

    def getLocation(self):
        return "%s:%d" % (self.uri, self.index)

    def __repr__(self):
        return "RecordMetaData(%s)" % self.getLocation()


class Record:
    def __init__(self, record, metaData=None):
        self.record, self.metaData = record, metaData

    def getRecord(self):
        return self.record

    def getMetaData(self):
        return self.metaData


class RecordReader:
    def __init__(self):
        self.split = None
        self._labels = None

    def initialize(self, split: InputSplit):
        self.split = split
        self.reset()
        return self

    def hasNext(self) -> bool:
        raise NotImplementedError

    def next(self) -> List[Writable]:
        raise NotImplementedError

    def nextRecord(self) -> Record:
        return Record(self.next(), None)

    def reset(self):
        raise NotImplementedError

    def resetSupported(self):
        return True

    def batchesSupported(self):
        return False

    def getLabels(self):
        return self._labels

    def close(self):
        pass

    def __iter__(self):
        self.reset()
        while self.hasNext():
            yield self.next()


class SequenceRecordReader(RecordReader):
    def sequenceRecord(self) -> List[List[Writable]]:
        raise NotImplementedError

    def next(self):
        return self.sequenceRecord()

    nextSequence = lambda self: Record(self.sequenceRecord(), None)


def _read_text(split: InputSplit):
    """Yield (uri, text) for every location of a split (StringSplit yields its payload)."""
    if isinstance(split, StringSplit):
        yield "<string>", split.data
        return
    for loc in split.locations():
        with open(loc, encoding="utf-8", errors="replace", newline="") as f:
            yield loc, f.read()


class LineRecordReader(RecordReader):
    def reset(self):
        self._lines = []
        if isinstance(self.split, ListStringSplit):
            self._lines = [("<list>", ",".join(r)) for r in self.split.data]
        else:
            for uri, txt in _read_text(self.split):
                self._lines += [(uri, l) for l in txt.splitlines()]
        self._i = 0

    def hasNext(self):
        return self._i < len(self._lines)

    def next(self):
        l = self._lines[self._i][1]
        self._i += 1
        return [Text(l)]

    def nextRecord(self):
        uri = self._lines[self._i][0]
        i = self._i
        return Record(self.next(), RecordMetaData(uri, i))


class CSVRecordReader(LineRecordReader):
    DEFAULT_DELIMITER, DEFAULT_QUOTE = ",", '"'

    def __init__(self, skipNumLines=0, delimiter=",", quote='"'):
        super().__init__()
        self.skip, self.delim, self.quote = int(skipNumLines), delimiter, quote

    def reset(self):
        self._rows = []
        if isinstance(self.split, ListStringSplit):
            self._rows = [("<list>", list(r)) for r in self.split.data]
        else:
            for uri, txt in _read_text(self.split):
                rd = csv.reader(io.StringIO(txt), delimiter=self.delim, quotechar=self.quote)
                for k, row in enumerate(rd):
                    if k >= self.skip and row:
                        self._rows.append((uri, row))
        self._i = 0

    def hasNext(self):
        return self._i < len(self._rows)

    def next(self):
        row = self._rows[self._i][1]
        self._i += 1
        return [Text(c) for c in row]

    def nextRecord(self):
        uri, i = self._rows[self._i][0], self._i
        return Record(self.next(), RecordMetaData(uri, i))

    def batchesSupported(self):
        return True

    def loadAllNumeric(self) -> torch.Tensor:
        """Whole split as one float32 [rows, cols] matrix through the native parser (falls back to numpy for quoted fields)."""
        from ..native import lib as L
        mats = []
        for loc in self.split.locations():
            m = None
            try:
                rt = L.rt()
                h = rt.dv_csv_parse(loc.encode(), self.delim.encode()[:1], self.skip)
                if h:
                    r, c = rt.dv_rows(h), rt.dv_cols(h)
                    m = torch.empty(r, c, dtype=torch.float32)
                    rt.dv_copy(h, m.data_ptr())
                    rt.dv_free(h)
            except Exception:
                m = None
            if m is None:
                m = torch.from_numpy(np.loadtxt(loc, delimiter=self.delim, skiprows=self.skip, dtype=np.float32, ndmin=2))
            mats.append(m)
        return torch.cat(mats, 0) if len(mats) != 1 else mats[0]


class CSVSequenceRecordReader(SequenceRecordReader):
    """One sequence per file; each line is a time step (CSVSequenceRecordReader.java)."""

    def __init__(self, skipNumLines=0, delimiter=","):
        super().__init__()
        self.skip, self.delim = int(skipNumLines), delimiter

    def reset(self):
        self._locs, self._i = list(self.split.locations()), 0

    def hasNext(self):
        return self._i < len(self._locs)

    def sequenceRecord(self):
        loc = self._locs[self._i]
        self._i += 1
        with open(loc, encoding="utf-8", newline="") as f:
            rows = [r for k, r in enumerate(csv.reader(f, delimiter=self.delim)) if k >= self.skip and r]
        return [[Text(c) for c in r] for r in rows]


class CSVRegexRecordReader(CSVRecordReader):
    """CSV where selected columns are further split by a regex with capture groups."""

    def __init__(self, skipNumLines=0, delimiter=",", quote='"', regexs: Optional[Sequence[Optional[str]]] = None):
        super().__init__(skipNumLines, delimiter, quote)
        self.regexs = [re.compile(r) if r else None for r in (regexs or [])]

    def next(self):
        row = super().next()
        out = []
        for k, w in enumerate(row):
            rx = self.regexs[k] if k < len(self.regexs) else None
            if rx is None:
                out.append(w)
            else:
                m = rx.fullmatch(w.toString())
                if not m:
                    raise ValueError("Invalid line: value does not match regex (%s) at column %d: %r" % (rx.pattern, k, w.toString()))
                out += [Text(g) for g in m.groups()]
        return out


class RegexLineRecordReader(LineRecordReader):
    def __init__(self, regex, skipNumLines=0):
        super().__init__()
        self.rx, self.skip = re.compile(regex), int(skipNumLines)

    def reset(self):
        super().reset()
        self._i = min(self.skip, len(self._lines))

    def next(self):
        line = super().next()[0].toString()
        m = self.rx.fullmatch(line)
        if not m:
            raise ValueError("Invalid line: line does not match regex (%s): %r" % (self.rx.pattern, line))
        return [Text(g) for g in m.groups()]


class SVMLightRecordReader(LineRecordReader):
    """``label idx:val idx:val ...`` -> [features..., label] (SVMLightRecordReader.java); indices 1-based unless zeroBasedIndexing."""

    def __init__(self, numFeatures=-1, zeroBasedIndexing=False, appendLabel=True, multilabel=False, numLabels=-1, zeroBasedLabelIndexing=False):
        super().__init__()
        self.numFeatures, self.zeroBased, self.appendLabel = int(numFeatures), zeroBasedIndexing, appendLabel
        self.multilabel, self.numLabels, self.zeroBasedLabel = multilabel, numLabels, zeroBasedLabelIndexing

    def reset(self):
        super().reset()
        self._lines = [(u, l) for u, l in self._lines if l.strip() and not l.lstrip().startswith("#")]

    def next(self):
        line = super().next()[0].toString().split("#")[0].split()
        feats = [0.0] * self.numFeatures
        for tok in line[1:]:
            if tok.startswith("qid:"):
                continue
            i, v = tok.split(":")
            k = int(i) - (0 if self.zeroBased else 1)
            if k < 0 or k >= self.numFeatures:
                raise IndexError("Found feature index %s outside [0, %d)" % (i, self.numFeatures))
            feats[k] = float(v)
        out = [DoubleWritable(v) for v in feats]
        if self.appendLabel:
            if self.multilabel:
                labs = [0.0] * self.numLabels
                if line[0]:
                    for l in line[0].split(","):
                        labs[int(l) - (0 if self.zeroBasedLabel else 1)] = 1.0
                out += [DoubleWritable(v) for v in labs]
            else:
                lab = float(line[0])
                out.append(IntWritable(int(lab)) if lab == int(lab) else DoubleWritable(lab))
        return out

    def loadAllNumeric(self) -> torch.Tensor:
        from ..native import lib as L
        rt = L.rt()
        mats = []
        for loc in self.split.locations():
            h = rt.dv_svmlight_parse(loc.encode(), self.numFeatures, 1 if self.zeroBased else 0)
            if not h:
                raise IOError(L.rt_last_error() if hasattr(L, "rt_last_error") else "svmlight parse failed: " + loc)
            m = torch.empty(rt.dv_rows(h), rt.dv_cols(h), dtype=torch.float32)
            rt.dv_copy(h, m.data_ptr())
            rt.dv_free(h)
            mats.append(torch.cat([m[:, 1:], m[:, :1]], 1))       # native layout is [label, features...]; records are [features..., label]
        return torch.cat(mats, 0) if len(mats) != 1 else mats[0]


LibSvmRecordReader = SVMLightRecordReader


class LibSvmRecordReader(SVMLightRecordReader):
    """LibSVM files share the SVMLight line format (reference: datavec-api records/reader/impl/misc/LibSvmRecordReader.java)."""


class JacksonLineRecordReader(LineRecordReader):
    """One JSON object per line; ``fieldSelection`` = list of dotted paths (with optional default values)."""

    def __init__(self, fieldSelection: Sequence, defaults: Optional[Sequence] = None):
        super().__init__()
        self.paths = [p.split(".") if isinstance(p, str) else list(p) for p in fieldSelection]
        self.defaults = list(defaults) if defaults is not None else [None] * len(self.paths)

    def next(self):
        obj = json.loads(super().next()[0].toString())
        out = []
        for path, dflt in zip(self.paths, self.defaults):
            cur = obj
            for k in path:
                cur = cur.get(k) if isinstance(cur, dict) else None
                if cur is None:
                    break
            out.append(wrap(cur if cur is not None else dflt))
        return out


class CollectionRecordReader(RecordReader):
    def __init__(self, records: Iterable[Sequence]):
        super().__init__()
        self._recs = [[wrap(v) for v in r] for r in records]
        self._i = 0

    def initialize(self, split=None):
        self.reset()
        return self

    def reset(self):
        self._i = 0

    def hasNext(self):
        return self._i < len(self._recs)

    def next(self):
        r = self._recs[self._i]
        self._i += 1
        return list(r)


InMemoryRecordReader = CollectionRecordReader


class CollectionSequenceRecordReader(SequenceRecordReader):
    def __init__(self, sequences: Iterable[Sequence[Sequence]]):
        super().__init__()
        self._seqs = [[[wrap(v) for v in step] for step in s] for s in sequences]
        self._i = 0

    def initialize(self, split=None):
        self.reset()
        return self

    def reset(self):
        self._i = 0

    def hasNext(self):
        return self._i < len(self._seqs)

    def sequenceRecord(self):
        s = self._seqs[self._i]
        self._i += 1
        return [list(step) for step in s]


InMemorySequenceRecordReader = CollectionSequenceRecordReader


class FileRecordReader(RecordReader):
    """Whole file content as one Text record; optional parent-directory label appended."""

    def __init__(self, appendLabel=False):
        super().__init__()
        self.appendLabel = appendLabel

    def reset(self):
        self._locs, self._i = list(self.split.locations()), 0
        if self.appendLabel:
            self._labels = sorted({os.path.basename(os.path.dirname(l)) for l in self._locs})

    def hasNext(self):
        return self._i < len(self._locs)

    def next(self):
        loc = self._locs[self._i]
        self._i += 1
        with open(loc, encoding="utf-8", errors="replace") as f:
            out = [Text(f.read())]
        if self.appendLabel:
            out.append(IntWritable(self._labels.index(os.path.basename(os.path.dirname(loc)))))
        return out


class TransformProcessRecordReader(RecordReader):
    def __init__(self, reader: RecordReader, transformProcess):
        super().__init__()
        self.reader, self.tp = reader, transformProcess
        self._next = None

    def initialize(self, split):
        self.reader.initialize(split)
        self._next = None
        return self

    def reset(self):
        self.reader.reset()
        self._next = None

    def _advance(self):
        while self._next is None and self.reader.hasNext():
            self._next = self.tp.execute(self.reader.next())          # None when filtered out

    def hasNext(self):
        self._advance()
        return self._next is not None

    def next(self):
        self._advance()
        r, self._next = self._next, None
        return r


class ConcatenatingRecordReader(RecordReader):
    def __init__(self, *readers):
        super().__init__()
        self.readers = readers
        self._k = 0

    def initialize(self, split=None):
        self.reset()
        return self

    def reset(self):
        for r in self.readers:
            r.reset()
        self._k = 0

    def hasNext(self):
        while self._k < len(self.readers) and not self.readers[self._k].hasNext():
            self._k += 1
        return self._k < len(self.readers)

    def next(self):
        self.hasNext()
        return self.readers[self._k].next()
