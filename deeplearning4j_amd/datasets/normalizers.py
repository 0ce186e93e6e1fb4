"""Data normalizers + binary serializer.

Reference: nd4j-api/.../linalg/dataset/api/preprocessor/{NormalizerStandardize, NormalizerMinMaxScaler,
ImagePreProcessingScaler, VGG16ImagePreProcessor, MultiNormalizerStandardize, MultiNormalizerMinMaxScaler,
MultiNormalizerHybrid}.java and serializer/NormalizerSerializer.java (header ``UTF("NORMALIZER") int(1)
UTF(type)`` followed by the per-type strategy payload written with Nd4j.write).

Statistics are accumulated with running (count, mean, M2) merges (Chan et al.), the same scheme as
DistributionStats.Builder, so fit(iterator) is one pass and exact regardless of batch size.
Time-series [mb, n, T] and image [mb, c, h, w] features are reduced over every axis but the channel
axis (stats/DistributionStats.java).
"""
from __future__ import annotations

import io
import struct
from typing import List, Optional

import torch

from ..nd4j import serde
from .dataset import DataSet, MultiDataSet


def _chan_view(t):
    """[mb, n, ...] -> [rows, n] view used for statistics."""
    if t.dim() <= 2:
        return t.reshape(t.shape[0], -1)
    perm = [0] + list(range(2, t.dim())) + [1]
    return t.permute(*perm).reshape(-1, t.shape[1])


def _bshape(t, v):
    """reshape a [n] statistic so it broadcasts along axis 1 of t."""
    if t.dim() <= 2:
        return v.reshape(1, -1)
    return v.reshape([1, -1] + [1] * (t.dim() - 2))


class _RunningStats:
    def __init__(self):
        self.n = 0
        self.mean = self.m2 = self.mn = self.mx = None

    def add(self, t, mask=None):
        x = _chan_view(t).double()
        if mask is not None and t.dim() == 3:
            keep = mask.reshape(-1) > 0
            x = x[keep.to(x.device)]
        x = x.cpu()
        k = x.shape[0]
        if k == 0:
            return
        mean = x.mean(0)
        m2 = ((x - mean) ** 2).sum(0)
        mn, mx = x.amin(0), x.amax(0)
        if self.n == 0:
            self.n, self.mean, self.m2, self.mn, self.mx = k, mean, m2, mn, mx
            return
        tot = self.n + k
        d = mean - self.mean
        self.m2 = self.m2 + m2 + d * d * (self.n * k / tot)
        self.mean = self.mean + d * (k / tot)
        self.mn, self.mx = torch.minimum(self.mn, mn), torch.maximum(self.mx, mx)
        self.n = tot

    def std(self):
        # population std (DistributionStats.Builder: sqrt(M2/n)); zero variance is guarded at transform time
        return torch.sqrt(self.m2 / max(self.n, 1))


class DistributionStats:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def getMean(self):
        return self.mean

    def getStd(self):
        return self.std


class MinMaxStats:
    def __init__(self, lower, upper):
        self.lower, self.upper = lower, upper

    def getLower(self):
        return self.lower

    def getUpper(self):
        return self.upper

    def getRange(self):
        r = self.upper - self.lower
        return torch.where(r == 0, torch.full_like(r, 1e-12), r)


class DataNormalization:
    TYPE = "CUSTOM"

    def __init__(self):
        self._fitLabel = False

    def fitLabel(self, flag=True):
        self._fitLabel = bool(flag)
        return self

    def isFitLabel(self):
        return self._fitLabel

    def getType(self):
        return self.TYPE

    # DataSetPreProcessor contract
    def preProcess(self, ds):
        self.transform(ds)

    def __call__(self, ds):
        self.preProcess(ds)
        return ds

    def _iter(self, data):
        if isinstance(data, (DataSet, MultiDataSet)):
            yield data
            return
        if data.resetSupported():
            data.reset()
        while data.hasNext():
            yield data.next()
        if data.resetSupported():
            data.reset()


# ---------------------------------------------------------------------------------- single-array normalizers
class NormalizerStandardize(DataNormalization):
    TYPE = "STANDARDIZE"

    def __init__(self, mean=None, std=None, labelMean=None, labelStd=None):
        super().__init__()
        self.featureStats = DistributionStats(mean, std) if mean is not None else None
        self.labelStats = DistributionStats(labelMean, labelStd) if labelMean is not None else None
        if labelMean is not None:
            self._fitLabel = True

    def fit(self, data):
        fs, ls = _RunningStats(), _RunningStats()
        for ds in self._iter(data):
            fs.add(ds.getFeatures(), ds.getFeaturesMaskArray())
            if self._fitLabel:
                ls.add(ds.getLabels(), ds.getLabelsMaskArray())
        self.featureStats = DistributionStats(fs.mean.float(), fs.std().float())
        if self._fitLabel:
            self.labelStats = DistributionStats(ls.mean.float(), ls.std().float())
        return self

    def getMean(self):
        return self.featureStats.mean

    def getStd(self):
        return self.featureStats.std

    def getLabelMean(self):
        return self.labelStats.mean

    def getLabelStd(self):
        return self.labelStats.std

    def setLabelStats(self, mean, std):
        self.labelStats = DistributionStats(mean, std)

    @staticmethod
    def _fwd(t, st):
        std = torch.where(st.std == 0, torch.ones_like(st.std), st.std).to(t.device, t.dtype if t.is_floating_point() else torch.float32)
        return (t - _bshape(t, st.mean.to(t.device, std.dtype))) / _bshape(t, std)

    @staticmethod
    def _rev(t, st):
        return t * _bshape(t, st.std.to(t.device, t.dtype)) + _bshape(t, st.mean.to(t.device, t.dtype))

    def transform(self, ds):
        if isinstance(ds, torch.Tensor):
            return self._fwd(ds, self.featureStats)
        ds.setFeatures(self._fwd(ds.getFeatures(), self.featureStats))
        if self._fitLabel and ds.getLabels() is not None:
            ds.setLabels(self._fwd(ds.getLabels(), self.labelStats))
        return ds

    def transformLabel(self, labels):
        return self._fwd(labels, self.labelStats) if self._fitLabel else labels

    def revert(self, ds):
        ds.setFeatures(self._rev(ds.getFeatures(), self.featureStats))
        if self._fitLabel and ds.getLabels() is not None:
            ds.setLabels(self._rev(ds.getLabels(), self.labelStats))
        return ds

    def revertFeatures(self, f):
        return self._rev(f, self.featureStats)

    def revertLabels(self, l):
        return self._rev(l, self.labelStats) if self._fitLabel else l


class NormalizerMinMaxScaler(DataNormalization):
    TYPE = "MIN_MAX"

    def __init__(self, minRange=0.0, maxRange=1.0):
        super().__init__()
        self.minRange, self.maxRange = float(minRange), float(maxRange)
        self.featureStats: Optional[MinMaxStats] = None
        self.labelStats: Optional[MinMaxStats] = None

    def getTargetMin(self):
        return self.minRange

    def getTargetMax(self):
        return self.maxRange

    def fit(self, data):
        fs, ls = _RunningStats(), _RunningStats()
        for ds in self._iter(data):
            fs.add(ds.getFeatures(), ds.getFeaturesMaskArray())
            if self._fitLabel:
                ls.add(ds.getLabels(), ds.getLabelsMaskArray())
        self.featureStats = MinMaxStats(fs.mn.float(), fs.mx.float())
        if self._fitLabel:
            self.labelStats = MinMaxStats(ls.mn.float(), ls.mx.float())
        return self

    def getMin(self):
        return self.featureStats.lower

    def getMax(self):
        return self.featureStats.upper

    def getLabelMin(self):
        return self.labelStats.lower

    def getLabelMax(self):
        return self.labelStats.upper

    def setFeatureStats(self, lo, hi):
        self.featureStats = MinMaxStats(lo, hi)

    def setLabelStats(self, lo, hi):
        self.labelStats = MinMaxStats(lo, hi)

    def _fwd(self, t, st):
        lo = _bshape(t, st.lower.to(t.device, t.dtype))
        rng = _bshape(t, st.getRange().to(t.device, t.dtype))
        return (t - lo) / rng * (self.maxRange - self.minRange) + self.minRange

    def _rev(self, t, st):
        lo = _bshape(t, st.lower.to(t.device, t.dtype))
        rng = _bshape(t, st.getRange().to(t.device, t.dtype))
        return (t - self.minRange) / (self.maxRange - self.minRange) * rng + lo

    def transform(self, ds):
        if isinstance(ds, torch.Tensor):
            return self._fwd(ds, self.featureStats)
        ds.setFeatures(self._fwd(ds.getFeatures(), self.featureStats))
        if self._fitLabel and ds.getLabels() is not None:
            ds.setLabels(self._fwd(ds.getLabels(), self.labelStats))
        return ds

    def revert(self, ds):
        ds.setFeatures(self._rev(ds.getFeatures(), self.featureStats))
        if self._fitLabel and ds.getLabels() is not None:
            ds.setLabels(self._rev(ds.getLabels(), self.labelStats))
        return ds

    def revertFeatures(self, f):
        return self._rev(f, self.featureStats)

    def revertLabels(self, l):
        return self._rev(l, self.labelStats) if self._fitLabel else l


class ImagePreProcessingScaler(DataNormalization):
    """pixel / maxPixelVal * (b - a) + a; no fitting needed (ImagePreProcessingScaler.java)."""
    TYPE = "IMAGE_MIN_MAX"

    def __init__(self, a=0.0, b=1.0, maxBits=8):
        super().__init__()
        self.minRange, self.maxRange = float(a), float(b)
        self.maxPixelVal = float(2 ** maxBits - 1)

    def setMaxPixelVal(self, v):
        self.maxPixelVal = float(v)

    def fit(self, data):
        return self

    def _fwd(self, t):
        t = t.float() if not t.is_floating_point() else t
        t = t / self.maxPixelVal
        if self.maxRange - self.minRange != 1.0:
            t = t * (self.maxRange - self.minRange)
        if self.minRange != 0.0:
            t = t + self.minRange
        return t

    def transform(self, ds):
        if isinstance(ds, torch.Tensor):
            return self._fwd(ds)
        ds.setFeatures(self._fwd(ds.getFeatures()))
        return ds

    def revertFeatures(self, f):
        return (f - self.minRange) / (self.maxRange - self.minRange) * self.maxPixelVal

    def revert(self, ds):
        ds.setFeatures(self.revertFeatures(ds.getFeatures()))
        return ds


class VGG16ImagePreProcessor(DataNormalization):
    """Subtract the ImageNet BGR channel means (VGG16ImagePreProcessor.java:42)."""
    TYPE = "IMAGE_VGG16"
    VGG_MEAN_OFFSET_BGR = (123.68, 116.779, 103.939)

    def fit(self, data):
        return self

    def _mean(self, t):
        return torch.tensor(self.VGG_MEAN_OFFSET_BGR, device=t.device, dtype=t.dtype).reshape(1, 3, 1, 1)

    def transform(self, ds):
        if isinstance(ds, torch.Tensor):
            return ds - self._mean(ds)
        f = ds.getFeatures().float()
        ds.setFeatures(f - self._mean(f))
        return ds

    def revertFeatures(self, f):
        return f + self._mean(f)

    def revert(self, ds):
        ds.setFeatures(self.revertFeatures(ds.getFeatures()))
        return ds


# ---------------------------------------------------------------------------------- MultiDataSet normalizers
class _MultiBase(DataNormalization):
    def numInputs(self):
        return len(self.featureStats)

    def numOutputs(self):
        return len(self.labelStats) if self.labelStats else 0

    def _collect(self, data):
        fs = ls = None
        for mds in self._iter(data):
            if fs is None:
                fs = [_RunningStats() for _ in range(mds.numFeatureArrays())]
                ls = [_RunningStats() for _ in range(mds.numLabelsArrays())]
            for i, s in enumerate(fs):
                s.add(mds.getFeatures(i), mds.getFeaturesMaskArray(i))
            if self._fitLabel:
                for i, s in enumerate(ls):
                    s.add(mds.getLabels(i), mds.getLabelsMaskArray(i))
        return fs, ls


class MultiNormalizerStandardize(_MultiBase):
    TYPE = "MULTI_STANDARDIZE"

    def __init__(self):
        super().__init__()
        self.featureStats: List[DistributionStats] = []
        self.labelStats: List[DistributionStats] = []

    def fit(self, data):
        fs, ls = self._collect(data)
        self.featureStats = [DistributionStats(s.mean.float(), s.std().float()) for s in fs]
        self.labelStats = [DistributionStats(s.mean.float(), s.std().float()) for s in ls] if self._fitLabel else []
        return self

    def setFeatureStats(self, st):
        self.featureStats = list(st)

    def setLabelStats(self, st):
        self.labelStats = list(st)

    def getFeatureMean(self, i):
        return self.featureStats[i].mean

    def getFeatureStd(self, i):
        return self.featureStats[i].std

    def getLabelMean(self, i):
        return self.labelStats[i].mean

    def getLabelStd(self, i):
        return self.labelStats[i].std

    def transform(self, mds):
        mds.features = [NormalizerStandardize._fwd(f, st) for f, st in zip(mds.features, self.featureStats)]
        if self._fitLabel:
            mds.labels = [NormalizerStandardize._fwd(l, st) for l, st in zip(mds.labels, self.labelStats)]
        return mds

    def revert(self, mds):
        mds.features = [NormalizerStandardize._rev(f, st) for f, st in zip(mds.features, self.featureStats)]
        if self._fitLabel:
            mds.labels = [NormalizerStandardize._rev(l, st) for l, st in zip(mds.labels, self.labelStats)]
        return mds


class MultiNormalizerMinMaxScaler(_MultiBase):
    TYPE = "MULTI_MIN_MAX"

    def __init__(self, minRange=0.0, maxRange=1.0):
        super().__init__()
        self.minRange, self.maxRange = float(minRange), float(maxRange)
        self._h = NormalizerMinMaxScaler(minRange, maxRange)
        self.featureStats: List[MinMaxStats] = []
        self.labelStats: List[MinMaxStats] = []

    def getTargetMin(self):
        return self.minRange

    def getTargetMax(self):
        return self.maxRange

    def fit(self, data):
        fs, ls = self._collect(data)
        self.featureStats = [MinMaxStats(s.mn.float(), s.mx.float()) for s in fs]
        self.labelStats = [MinMaxStats(s.mn.float(), s.mx.float()) for s in ls] if self._fitLabel else []
        return self

    def setFeatureStats(self, st):
        self.featureStats = list(st)

    def setLabelStats(self, st):
        self.labelStats = list(st)

    def getMin(self, i):
        return self.featureStats[i].lower

    def getMax(self, i):
        return self.featureStats[i].upper

    def getLabelMin(self, i):
        return self.labelStats[i].lower

    def getLabelMax(self, i):
        return self.labelStats[i].upper

    def transform(self, mds):
        mds.features = [self._h._fwd(f, st) for f, st in zip(mds.features, self.featureStats)]
        if self._fitLabel:
            mds.labels = [self._h._fwd(l, st) for l, st in zip(mds.labels, self.labelStats)]
        return mds

    def revert(self, mds):
        mds.features = [self._h._rev(f, st) for f, st in zip(mds.features, self.featureStats)]
        if self._fitLabel:
            mds.labels = [self._h._rev(l, st) for l, st in zip(mds.labels, self.labelStats)]
        return mds


class MultiNormalizerHybrid(_MultiBase):
    """Per-input / per-output choice of strategy (MultiNormalizerHybrid.java)."""
    TYPE = "MULTI_HYBRID"

    def __init__(self):
        super().__init__()
        self.inputStrategy, self.outputStrategy = {}, {}
        self.globalInput = self.globalOutput = None
        self.featureStats, self.labelStats = {}, {}

    def standardizeAllInputs(self):
        self.globalInput = ("std",)
        return self

    def minMaxScaleAllInputs(self, lo=0.0, hi=1.0):
        self.globalInput = ("minmax", lo, hi)
        return self

    def standardizeInput(self, i):
        self.inputStrategy[i] = ("std",)
        return self

    def minMaxScaleInput(self, i, lo=0.0, hi=1.0):
        self.inputStrategy[i] = ("minmax", lo, hi)
        return self

    def standardizeAllOutputs(self):
        self.globalOutput = ("std",)
        return self

    def minMaxScaleAllOutputs(self, lo=0.0, hi=1.0):
        self.globalOutput = ("minmax", lo, hi)
        return self

    def standardizeOutput(self, i):
        self.outputStrategy[i] = ("std",)
        return self

    def minMaxScaleOutput(self, i, lo=0.0, hi=1.0):
        self.outputStrategy[i] = ("minmax", lo, hi)
        return self

    def _strat(self, i, per, glob):
        return per.get(i, glob)

    def fit(self, data):
        self._fitLabel = True
        fs, ls = self._collect(data)
        mk = lambda s, st: DistributionStats(s.mean.float(), s.std().float()) if st[0] == "std" else MinMaxStats(s.mn.float(), s.mx.float())
        self.featureStats = {i: mk(s, self._strat(i, self.inputStrategy, self.globalInput)) for i, s in enumerate(fs)
                             if self._strat(i, self.inputStrategy, self.globalInput)}
        self.labelStats = {i: mk(s, self._strat(i, self.outputStrategy, self.globalOutput)) for i, s in enumerate(ls)
                           if self._strat(i, self.outputStrategy, self.globalOutput)}
        return self

    def _apply(self, arrs, stats, per, glob, rev):
        out = list(arrs)
        for i, st in stats.items():
            strat = self._strat(i, per, glob)
            if strat[0] == "std":
                out[i] = (NormalizerStandardize._rev if rev else NormalizerStandardize._fwd)(arrs[i], st)
            else:
                h = NormalizerMinMaxScaler(strat[1], strat[2])
                out[i] = (h._rev if rev else h._fwd)(arrs[i], st)
        return out

    def transform(self, mds):
        mds.features = self._apply(mds.features, self.featureStats, self.inputStrategy, self.globalInput, False)
        mds.labels = self._apply(mds.labels, self.labelStats, self.outputStrategy, self.globalOutput, False)
        return mds

    def revert(self, mds):
        mds.features = self._apply(mds.features, self.featureStats, self.inputStrategy, self.globalInput, True)
        mds.labels = self._apply(mds.labels, self.labelStats, self.outputStrategy, self.globalOutput, True)
        return mds


# ---------------------------------------------------------------------------------- serializer
def _w(out, t):
    serde.write_array(t.reshape(1, -1).float(), out)


def _r(inp):
    return serde.read_array(inp).reshape(-1).float()


class NormalizerSerializer:
    HEADER = "NORMALIZER"
    _default = None

    def __init__(self):
        self._custom = {}

    @classmethod
    def getDefault(cls):
        if cls._default is None:
            cls._default = cls()
        return cls._default

    def addStrategy(self, name, writer, reader):
        """Custom strategies: ``writer(normalizer, stream)``, ``reader(stream) -> normalizer``."""
        self._custom[name] = (writer, reader)
        return self

    def write(self, n, target):
        if isinstance(target, (str, bytes)) or hasattr(target, "__fspath__"):
            with open(target, "wb") as f:
                return self.write(n, f)
        out = target
        serde._write_utf(out, self.HEADER)
        out.write(struct.pack(">i", 1))
        t = n.getType()
        serde._write_utf(out, t)
        if t == "STANDARDIZE":
            out.write(struct.pack(">?", n.isFitLabel()))
            _w(out, n.getMean()), _w(out, n.getStd())
            if n.isFitLabel():
                _w(out, n.getLabelMean()), _w(out, n.getLabelStd())
        elif t == "MIN_MAX":
            out.write(struct.pack(">?dd", n.isFitLabel(), n.getTargetMin(), n.getTargetMax()))
            _w(out, n.getMin()), _w(out, n.getMax())
            if n.isFitLabel():
                _w(out, n.getLabelMin()), _w(out, n.getLabelMax())
        elif t == "IMAGE_MIN_MAX":
            out.write(struct.pack(">ddd", n.minRange, n.maxRange, n.maxPixelVal))
        elif t == "IMAGE_VGG16":
            pass
        elif t == "MULTI_STANDARDIZE":
            out.write(struct.pack(">?ii", n.isFitLabel(), n.numInputs(), n.numOutputs() if n.isFitLabel() else -1))
            for s in n.featureStats:
                _w(out, s.mean), _w(out, s.std)
            if n.isFitLabel():
                for s in n.labelStats:
                    _w(out, s.mean), _w(out, s.std)
        elif t == "MULTI_MIN_MAX":
            out.write(struct.pack(">?ii", n.isFitLabel(), n.numInputs(), n.numOutputs() if n.isFitLabel() else -1))
            out.write(struct.pack(">dd", n.getTargetMin(), n.getTargetMax()))
            for s in n.featureStats:
                _w(out, s.lower), _w(out, s.upper)
            if n.isFitLabel():
                for s in n.labelStats:
                    _w(out, s.lower), _w(out, s.upper)
        elif t == "CUSTOM":
            name = type(n).__name__
            if name not in self._custom:
                raise RuntimeError("No serializer strategy registered for custom normalizer " + name)
            serde._write_utf(out, name)
            self._custom[name][0](n, out)
        else:
            raise RuntimeError("No serializer strategy for normalizer type " + t)

    def restore(self, source):
        if isinstance(source, (str, bytes)) or hasattr(source, "__fspath__"):
            with open(source, "rb") as f:
                return self.restore(io.BytesIO(f.read()))
        inp = source
        if serde._read_utf(inp) != self.HEADER:
            raise ValueError("Could not restore normalizer: invalid header")
        (ver,) = struct.unpack(">i", inp.read(4))
        if ver != 1:
            raise ValueError("Could not restore normalizer: invalid version (%d)" % ver)
        t = serde._read_utf(inp)
        if t == "STANDARDIZE":
            (fl,) = struct.unpack(">?", inp.read(1))
            n = NormalizerStandardize(_r(inp), _r(inp))
            n.fitLabel(fl)
            if fl:
                n.setLabelStats(_r(inp), _r(inp))
            return n
        if t == "MIN_MAX":
            fl, lo, hi = struct.unpack(">?dd", inp.read(17))
            n = NormalizerMinMaxScaler(lo, hi)
            n.fitLabel(fl)
            n.setFeatureStats(_r(inp), _r(inp))
            if fl:
                n.setLabelStats(_r(inp), _r(inp))
            return n
        if t == "IMAGE_MIN_MAX":
            a, b, mp = struct.unpack(">ddd", inp.read(24))
            n = ImagePreProcessingScaler(a, b)
            n.setMaxPixelVal(mp)
            return n
        if t == "IMAGE_VGG16":
            return VGG16ImagePreProcessor()
        if t in ("MULTI_STANDARDIZE", "MULTI_MIN_MAX"):
            fl, ni, no = struct.unpack(">?ii", inp.read(9))
            if t == "MULTI_STANDARDIZE":
                n, mk = MultiNormalizerStandardize(), DistributionStats
            else:
                lo, hi = struct.unpack(">dd", inp.read(16))
                n, mk = MultiNormalizerMinMaxScaler(lo, hi), MinMaxStats
            n.fitLabel(fl)
            n.setFeatureStats([mk(_r(inp), _r(inp)) for _ in range(ni)])
            if fl:
                n.setLabelStats([mk(_r(inp), _r(inp)) for _ in range(no)])
            return n
        if t == "CUSTOM":
            name = serde._read_utf(inp)
            if name not in self._custom:
                raise RuntimeError("No serializer strategy registered for custom normalizer " + name)
            return self._custom[name][1](inp)
        raise RuntimeError("No serializer strategy found for header type " + t)
