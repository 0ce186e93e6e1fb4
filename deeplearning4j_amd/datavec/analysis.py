"""DataVec analysis / quality / HTML report / sorted rank.

Reference (behaviour, not code), all under datavec/datavec-api/src/main/java/org/datavec/api/transform/:
* analysis/columns/*Analysis.java, analysis/DataAnalysis.java, analysis/SequenceDataAnalysis.java,
  analysis/sequence/SequenceLengthAnalysis.java, analysis/counter/*Counter.java, analysis/histogram/*HistogramCounter.java
* quality/DataQualityAnalysis.java, quality/columns/*Quality.java, analysis/quality/**/*QualityAddFunction.java
* ui/HtmlAnalysis.java, ui/HtmlSequencePlotting.java
* rank/CalculateSortedRank.java (+ datavec-local LocalTransformExecutor.java:469-500 for its execution)
* datavec-local AnalyzeLocal.java

The counters of the reference fold one Writable at a time through add / merge functions (Spark aggregate shape). Here a column is
reduced in ONE vectorised pass over a NumPy array (the t-digest becomes exact quantiles; HyperLogLog becomes an exact set), which is
what a single-node loader feeding a 288 GB GPU wants: the analysis of a numeric column is a handful of array reductions, not a
per-record virtual call chain.
"""
from __future__ import annotations

import html
import json
import math
import re
from typing import Dict, List, Optional, Sequence

import numpy as np

from .transform import ColumnAnalysis, ColumnMetaData, ColumnType, DataAnalysis as _BaseDataAnalysis, Schema, _SortedRank
from .writable import (BytesWritable, DoubleWritable, FloatWritable, IntWritable, LongWritable, NDArrayWritable, NullWritable, Text,
                       Writable)

_NUMERIC = (ColumnType.Integer, ColumnType.Long, ColumnType.Double, ColumnType.Float, ColumnType.Time)


# ================================================================================================ column analyses
class NumericalColumnAnalysis(ColumnAnalysis):
    """Shared numeric statistics (NumericalColumnAnalysis.java): mean, sample stdev / variance, sign counts, histogram, quantiles
    (t-digest upstream, exact here)."""
    KIND = "Numerical"

    def __init__(self, type_):
        super().__init__(type_)
        self.sampleVariance = float("nan")
        self.countZero = self.countNegative = self.countPositive = self.countMinValue = self.countMaxValue = 0
        self.countNaN = 0
        self.histogramBucketCounts = None
        self._quantiles = None

    # reference getter names
    def getCountTotal(self): return self.count
    def getSampleVariance(self): return self.sampleVariance
    def getSampleStdev(self): return self.stdev
    def getCountZero(self): return self.countZero
    def getCountNegative(self): return self.countNegative
    def getCountPositive(self): return self.countPositive
    def getCountMinValue(self): return self.countMinValue
    def getCountMaxValue(self): return self.countMaxValue
    def getHistogramBuckets(self): return self.histogramBuckets
    def getHistogramBucketCounts(self): return self.histogramBucketCounts
    def getMinDouble(self): return float(self.min)
    def getMaxDouble(self): return float(self.max)

    def getQuantile(self, q: float) -> float:
        if self._quantiles is None:
            return float("nan")
        qs, vs = self._quantiles
        return float(np.interp(q, qs, vs))

    def _fill(self, v: np.ndarray, maxBuckets: int, integer: bool):
        self.countNaN = int(np.isnan(v).sum()) if v.dtype.kind == "f" else 0
        v = v[~np.isnan(v)] if v.dtype.kind == "f" else v
        if v.size == 0:
            return
        vmin, vmax = v.min(), v.max()
        self.min, self.max = (int(vmin), int(vmax)) if integer else (float(vmin), float(vmax))
        self.mean = float(v.mean())
        self.sampleVariance = float(v.var(ddof=1)) if v.size > 1 else 0.0
        self.stdev = math.sqrt(self.sampleVariance)
        self.countZero = int((v == 0).sum())
        self.countNegative = int((v < 0).sum())
        self.countPositive = int((v > 0).sum())
        self.countMinValue = int((v == vmin).sum())
        self.countMaxValue = int((v == vmax).sum())
        self.uniqueCount = int(np.unique(v).size)
        nb = max(1, min(int(maxBuckets), self.uniqueCount))
        cnt, edges = np.histogram(v.astype(np.float64), bins=nb)
        self.histogramBuckets, self.histogramBucketCounts = edges.tolist(), cnt.tolist()
        self.histogramCounts = self.histogramBucketCounts
        qs = np.linspace(0.0, 1.0, 101)
        self._quantiles = (qs, np.quantile(v.astype(np.float64), qs))

    def to_dict(self):
        d = {"@class": self.KIND + "Analysis", "columnType": self.type, "countTotal": self.count, "countMissing": self.countMissing,
             "min": self.min, "max": self.max, "mean": self.mean, "sampleStdev": self.stdev, "sampleVariance": self.sampleVariance,
             "countZero": self.countZero, "countNegative": self.countNegative, "countPositive": self.countPositive,
             "countMinValue": self.countMinValue, "countMaxValue": self.countMaxValue, "countNaN": self.countNaN,
             "countUnique": self.uniqueCount, "histogramBuckets": self.histogramBuckets,
             "histogramBucketCounts": self.histogramBucketCounts}
        if self._quantiles is not None:
            d["quantiles"] = [float(x) for x in self._quantiles[1]]
        return d

    def __repr__(self):
        return ("%sAnalysis(min=%s,max=%s,mean=%s,sampleStDev=%s,sampleVariance=%s,countZero=%d,countNegative=%d,countPositive=%d,"
                "countMinValue=%d,countMaxValue=%d,count=%d%s)") % (
            self.KIND, self.min, self.max, self.mean, self.stdev, self.sampleVariance, self.countZero, self.countNegative,
            self.countPositive, self.countMinValue, self.countMaxValue, self.count,
            (",countNaN=%d" % self.countNaN) if self.KIND == "Double" else "")


class IntegerAnalysis(NumericalColumnAnalysis):
    KIND = "Integer"


class LongAnalysis(NumericalColumnAnalysis):
    KIND = "Long"


class DoubleAnalysis(NumericalColumnAnalysis):
    KIND = "Double"

    def getCountNaN(self): return self.countNaN


class TimeAnalysis(NumericalColumnAnalysis):
    KIND = "Time"


class StringAnalysis(ColumnAnalysis):
    """StringAnalysis.java: unique count and the length statistics (min / max / mean / sample stdev of the string lengths, with a
    histogram over lengths)."""

    def __init__(self):
        super().__init__(ColumnType.String)
        self.meanLength = self.sampleStdevLength = self.sampleVarianceLength = float("nan")

    def getCountUnique(self): return self.uniqueCount
    def getMinLength(self): return self.minLength
    def getMaxLength(self): return self.maxLength
    def getMeanLength(self): return self.meanLength
    def getSampleStdevLength(self): return self.sampleStdevLength
    def getCountTotal(self): return self.count

    def _fill(self, ss: List[str], maxBuckets: int):
        for s in ss:
            self.stateCounts[s] = self.stateCounts.get(s, 0) + 1
        self.uniqueCount = len(self.stateCounts)
        if ss:
            ln = np.fromiter(map(len, ss), dtype=np.int64, count=len(ss))
            self.minLength, self.maxLength = int(ln.min()), int(ln.max())
            self.meanLength = float(ln.mean())
            self.sampleVarianceLength = float(ln.var(ddof=1)) if ln.size > 1 else 0.0
            self.sampleStdevLength = math.sqrt(self.sampleVarianceLength)
            nb = max(1, min(int(maxBuckets), int(np.unique(ln).size)))
            cnt, edges = np.histogram(ln.astype(np.float64), bins=nb)
            self.histogramBuckets, self.histogramCounts = edges.tolist(), cnt.tolist()

    def to_dict(self):
        return {"@class": "StringAnalysis", "columnType": self.type, "countTotal": self.count, "countUnique": self.uniqueCount,
                "minLength": self.minLength, "maxLength": self.maxLength, "meanLength": self.meanLength,
                "sampleStdevLength": self.sampleStdevLength, "sampleVarianceLength": self.sampleVarianceLength,
                "histogramBuckets": self.histogramBuckets, "histogramBucketCounts": self.histogramCounts}

    def __repr__(self):
        return "StringAnalysis(unique=%d,minLen=%d,maxLen=%d,meanLen=%s,sampleStDevLen=%s,sampleVarianceLen=%s,count=%d)" % (
            self.uniqueCount, self.minLength, self.maxLength, self.meanLength, self.sampleStdevLength, self.sampleVarianceLength,
            self.count)


class CategoricalAnalysis(ColumnAnalysis):
    def __init__(self):
        super().__init__(ColumnType.Categorical)

    def getMapOfCounts(self): return dict(self.stateCounts)
    def getCountTotal(self): return self.count

    def to_dict(self):
        return {"@class": "CategoricalAnalysis", "columnType": self.type, "countTotal": self.count, "mapOfCounts": dict(self.stateCounts)}

    def __repr__(self):
        return "CategoricalAnalysis(CategoryCounts=%s)" % (self.stateCounts,)


class BytesAnalysis(ColumnAnalysis):
    def __init__(self):
        super().__init__(ColumnType.Bytes)
        self.countNull = 0
        self.countZeroLength = 0
        self.minNumBytes = self.maxNumBytes = 0

    def getCountTotal(self): return self.count

    def to_dict(self):
        return {"@class": "BytesAnalysis", "columnType": self.type, "countTotal": self.count, "countNull": self.countNull,
                "countZeroLength": self.countZeroLength, "minNumBytes": self.minNumBytes, "maxNumBytes": self.maxNumBytes}

    def __repr__(self):
        return "BytesAnalysis(countTotal=%d,countNull=%d,countZeroLength=%d,minNumBytes=%d,maxNumBytes=%d)" % (
            self.count, self.countNull, self.countZeroLength, self.minNumBytes, self.maxNumBytes)


class NDArrayAnalysis(ColumnAnalysis):
    """NDArrayAnalysis.java: counts, min / max array length, value range over all arrays, counts by rank."""

    def __init__(self):
        super().__init__(ColumnType.NDArray)
        self.countNull = 0
        self.minLength = self.maxLength = 0
        self.totalNDArrayValues = 0
        self.countsByRank: Dict[int, int] = {}
        self.minValue = self.maxValue = float("nan")

    def getCountTotal(self): return self.count

    def to_dict(self):
        return {"@class": "NDArrayAnalysis", "columnType": self.type, "countTotal": self.count, "countNull": self.countNull,
                "minLength": self.minLength, "maxLength": self.maxLength, "totalNDArrayValues": self.totalNDArrayValues,
                "countsByRank": {str(k): v for k, v in self.countsByRank.items()}, "minValue": self.minValue,
                "maxValue": self.maxValue, "histogramBuckets": self.histogramBuckets, "histogramBucketCounts": self.histogramCounts}

    def __repr__(self):
        return ("NDArrayAnalysis(countTotal=%d,countNull=%d,minLength=%d,maxLength=%d,totalValuesAllNDArrays=%d,minValue=%s,"
                "maxValue=%s,countsByNDArrayRank=%s)") % (self.count, self.countNull, self.minLength, self.maxLength,
                                                          self.totalNDArrayValues, self.minValue, self.maxValue, self.countsByRank)


def _is_missing(w) -> bool:
    return w is None or isinstance(w, NullWritable) or (isinstance(w, Text) and (w.toString() is None or w.toString() == ""))


def analyze_column(meta: ColumnMetaData, col: Sequence[Writable], maxHistogramBuckets: int = 20) -> ColumnAnalysis:
    t = meta.type
    if t in _NUMERIC:
        a = {ColumnType.Integer: IntegerAnalysis, ColumnType.Long: LongAnalysis, ColumnType.Time: TimeAnalysis}.get(t, DoubleAnalysis)(t)
        a.count = len(col)
        vals = []
        integer = t in (ColumnType.Integer, ColumnType.Long, ColumnType.Time)
        for w in col:
            if _is_missing(w):
                a.countMissing += 1
                continue
            try:
                vals.append(w.toDouble())
            except (ValueError, TypeError):
                a.countMissing += 1
        v = np.asarray(vals, dtype=np.float64)
        if integer and v.size and not np.isnan(v).any():
            v = v.astype(np.int64)
        a._fill(v, maxHistogramBuckets, integer)
        a.countMissing += a.countNaN              # a NaN carries no value: reported under countNaN AND as missing
        return a
    if t == ColumnType.Categorical:
        a = CategoricalAnalysis()
        a.count = len(col)
        for s in (meta.stateNames or []):
            a.stateCounts[s] = 0
        for w in col:
            s = w.toString()
            a.stateCounts[s] = a.stateCounts.get(s, 0) + 1
        a.uniqueCount = sum(1 for c in a.stateCounts.values() if c)
        return a
    if t == ColumnType.Bytes:
        a = BytesAnalysis()
        a.count = len(col)
        lens = []
        for w in col:
            b = getattr(w, "value", None) if isinstance(w, BytesWritable) else None
            if b is None:
                a.countNull += 1
            else:
                lens.append(len(b))
        a.countZeroLength = sum(1 for n in lens if n == 0)
        if lens:
            a.minNumBytes, a.maxNumBytes = min(lens), max(lens)
        return a
    if t == ColumnType.NDArray:
        a = NDArrayAnalysis()
        a.count = len(col)
        flat = []
        for w in col:
            arr = getattr(w, "value", None) if isinstance(w, NDArrayWritable) else None
            if arr is None:
                a.countNull += 1
                continue
            x = np.asarray(arr.detach().cpu().numpy() if hasattr(arr, "detach") else (arr.tensor().cpu().numpy() if hasattr(arr, "tensor") else arr))
            a.countsByRank[x.ndim] = a.countsByRank.get(x.ndim, 0) + 1
            flat.append(x.reshape(-1).astype(np.float64))
        if flat:
            lens = [f.size for f in flat]
            a.minLength, a.maxLength, a.totalNDArrayValues = min(lens), max(lens), int(sum(lens))
            allv = np.concatenate(flat) if a.totalNDArrayValues else np.zeros(0)
            if allv.size:
                a.minValue, a.maxValue = float(allv.min()), float(allv.max())
                cnt, edges = np.histogram(allv, bins=max(1, min(maxHistogramBuckets, int(np.unique(allv).size))))
                a.histogramBuckets, a.histogramCounts = edges.tolist(), cnt.tolist()
        return a
    a = StringAnalysis()
    a.type = t
    a.count = len(col)
    a._fill([w.toString() for w in col], maxHistogramBuckets)
    return a


class DataAnalysis(_BaseDataAnalysis):
    """DataAnalysis.java: per-column analyses + JSON / YAML round trip (toJson / fromJson / toYaml / fromYaml)."""

    def getColumnAnalysis(self, name_or_idx=None):
        if name_or_idx is None:
            return list(self.columnAnalysis)
        return super().getColumnAnalysis(name_or_idx)

    def to_dict(self):
        cols = []
        for c, a in zip(self.schema.columns, self.columnAnalysis):
            d = a.to_dict() if hasattr(a, "to_dict") else {"@class": "ColumnAnalysis", "columnType": a.type, "countTotal": a.count}
            cols.append({"columnName": c.name, "meta": c.to_json(), "analysis": d})
        return {"DataAnalysis": cols}

    def toJson(self) -> str:
        return json.dumps(self.to_dict(), indent=2, allow_nan=True)

    def toYaml(self) -> str:
        import yaml
        return yaml.safe_dump(json.loads(self.toJson()), sort_keys=False)

    @staticmethod
    def _from_dict(d) -> "DataAnalysis":
        metas, analyses = [], []
        for e in d["DataAnalysis"]:
            m = e["meta"]
            metas.append(ColumnMetaData(m["name"], m["type"], m.get("stateNames"), m.get("min"), m.get("max")))
            analyses.append(_analysis_from_dict(e["analysis"]))
        return DataAnalysis(Schema(metas), analyses)

    @staticmethod
    def fromJson(s: str) -> "DataAnalysis":
        return DataAnalysis._from_dict(json.loads(s))

    @staticmethod
    def fromYaml(s: str) -> "DataAnalysis":
        import yaml
        return DataAnalysis._from_dict(yaml.safe_load(s))

    def __str__(self):
        w = max([len(c.name) for c in self.schema.columns] + [4]) + 8
        lines = ["%-6s%-*s%-15s%s" % ("idx", w, "name", "type", "analysis")]
        for i, (c, a) in enumerate(zip(self.schema.columns, self.columnAnalysis)):
            lines.append("%-6d%-*s%-15s%s" % (i, w, '"%s"' % c.name, c.type, a))
        return "\n".join(lines)


def _analysis_from_dict(d) -> ColumnAnalysis:
    k = d.get("@class", "")
    if k in ("IntegerAnalysis", "LongAnalysis", "DoubleAnalysis", "TimeAnalysis"):
        a = {"IntegerAnalysis": IntegerAnalysis, "LongAnalysis": LongAnalysis, "DoubleAnalysis": DoubleAnalysis,
             "TimeAnalysis": TimeAnalysis}[k](d["columnType"])
        a.count, a.countMissing = d["countTotal"], d.get("countMissing", 0)
        a.min, a.max, a.mean, a.stdev, a.sampleVariance = d["min"], d["max"], d["mean"], d["sampleStdev"], d["sampleVariance"]
        for f in ("countZero", "countNegative", "countPositive", "countMinValue", "countMaxValue", "countNaN"):
            setattr(a, f, d.get(f, 0))
        a.uniqueCount = d.get("countUnique", 0)
        a.histogramBuckets, a.histogramBucketCounts = d.get("histogramBuckets"), d.get("histogramBucketCounts")
        a.histogramCounts = a.histogramBucketCounts
        if d.get("quantiles"):
            a._quantiles = (np.linspace(0.0, 1.0, len(d["quantiles"])), np.asarray(d["quantiles"], dtype=np.float64))
        return a
    if k == "StringAnalysis":
        a = StringAnalysis()
        a.type = d.get("columnType", ColumnType.String)
        a.count, a.uniqueCount = d["countTotal"], d["countUnique"]
        a.minLength, a.maxLength, a.meanLength = d["minLength"], d["maxLength"], d["meanLength"]
        a.sampleStdevLength, a.sampleVarianceLength = d["sampleStdevLength"], d["sampleVarianceLength"]
        a.histogramBuckets, a.histogramCounts = d.get("histogramBuckets"), d.get("histogramBucketCounts")
        return a
    if k == "CategoricalAnalysis":
        a = CategoricalAnalysis()
        a.count, a.stateCounts = d["countTotal"], dict(d["mapOfCounts"])
        a.uniqueCount = sum(1 for c in a.stateCounts.values() if c)
        return a
    if k == "BytesAnalysis":
        a = BytesAnalysis()
        a.count, a.countNull, a.countZeroLength = d["countTotal"], d["countNull"], d["countZeroLength"]
        a.minNumBytes, a.maxNumBytes = d["minNumBytes"], d["maxNumBytes"]
        return a
    if k == "NDArrayAnalysis":
        a = NDArrayAnalysis()
        a.count, a.countNull = d["countTotal"], d["countNull"]
        a.minLength, a.maxLength, a.totalNDArrayValues = d["minLength"], d["maxLength"], d["totalNDArrayValues"]
        a.countsByRank = {int(k2): v for k2, v in d["countsByRank"].items()}
        a.minValue, a.maxValue = d["minValue"], d["maxValue"]
        a.histogramBuckets, a.histogramCounts = d.get("histogramBuckets"), d.get("histogramBucketCounts")
        return a
    a = ColumnAnalysis(d.get("columnType"))
    a.count = d.get("countTotal", 0)
    return a


class SequenceLengthAnalysis:
    """SequenceLengthAnalysis.java: statistics over the lengths of the sequences of a data set."""

    def __init__(self, lengths: Sequence[int], maxHistogramBuckets: int = 20):
        ln = np.asarray(list(lengths), dtype=np.int64)
        self.totalNumSequences = int(ln.size)
        self.minSeqLength = int(ln.min()) if ln.size else 0
        self.maxSeqLength = int(ln.max()) if ln.size else 0
        self.countZeroLength = int((ln == 0).sum())
        self.countOneLength = int((ln == 1).sum())
        self.meanLength = float(ln.mean()) if ln.size else float("nan")
        if ln.size:
            cnt, edges = np.histogram(ln.astype(np.float64), bins=max(1, min(maxHistogramBuckets, int(np.unique(ln).size))))
            self.histogramBuckets, self.histogramBucketCounts = edges.tolist(), cnt.tolist()
        else:
            self.histogramBuckets = self.histogramBucketCounts = None

    def __repr__(self):
        return ("SequenceLengthAnalysis(totalNumSequences=%d,minSeqLength=%d,maxSeqLength=%d,countZeroLength=%d,countOneLength=%d,"
                "meanLength=%s)") % (self.totalNumSequences, self.minSeqLength, self.maxSeqLength, self.countZeroLength,
                                     self.countOneLength, self.meanLength)


class SequenceDataAnalysis(DataAnalysis):
    def __init__(self, schema, analyses, sequenceLengthAnalysis: SequenceLengthAnalysis):
        super().__init__(schema, analyses)
        self.sequenceLengthAnalysis = sequenceLengthAnalysis

    def getSequenceLengthAnalysis(self):
        return self.sequenceLengthAnalysis

    def __str__(self):
        return "%s\n%s" % (self.sequenceLengthAnalysis, super().__str__())


# ================================================================================================ quality
class ColumnQuality:
    """Base counters (ColumnQuality.java): valid / invalid / missing / total."""
    NAME = "ColumnQuality"
    EXTRA: Sequence[str] = ()

    def __init__(self, countValid=0, countInvalid=0, countMissing=0, countTotal=0, **extra):
        self.countValid, self.countInvalid, self.countMissing, self.countTotal = countValid, countInvalid, countMissing, countTotal
        for k in self.EXTRA:
            setattr(self, k, extra.get(k, 0))

    def getCountValid(self): return self.countValid
    def getCountInvalid(self): return self.countInvalid
    def getCountMissing(self): return self.countMissing
    def getCountTotal(self): return self.countTotal

    def add(self, other: "ColumnQuality") -> "ColumnQuality":
        if type(other) is not type(self):
            raise TypeError("cannot add %s to %s" % (type(other).__name__, type(self).__name__))
        kw = {k: getattr(self, k) + getattr(other, k) for k in self.EXTRA}
        return type(self)(self.countValid + other.countValid, self.countInvalid + other.countInvalid,
                          self.countMissing + other.countMissing, self.countTotal + other.countTotal, **kw)

    def _base(self):
        return "countValid=%d, countInvalid=%d, countMissing=%d, countTotal=%d" % (self.countValid, self.countInvalid, self.countMissing,
                                                                                 self.countTotal)

    def __repr__(self):
        extra = "".join(", %s=%s" % (k, getattr(self, k)) for k in self.EXTRA)
        return "%s(%s%s)" % (self.NAME, self._base(), extra)

    def __eq__(self, o):
        return type(o) is type(self) and repr(o) == repr(self)


class DoubleQuality(ColumnQuality):
    NAME, EXTRA = "DoubleQuality", ("countNonReal", "countNaN", "countInfinite")


class IntegerQuality(ColumnQuality):
    NAME, EXTRA = "IntegerQuality", ("countNonInteger",)


class LongQuality(ColumnQuality):
    NAME, EXTRA = "LongQuality", ("countNonLong",)


class TimeQuality(ColumnQuality):
    NAME = "TimeQuality"


class BytesQuality(ColumnQuality):
    NAME = "BytesQuality"


class CategoricalQuality(ColumnQuality):
    NAME = "CategoricalQuality"


class StringQuality(ColumnQuality):
    """StringQuality.java; the HyperLogLog sketch of the reference is an exact set here (`countApproxUnique` keeps its name)."""
    NAME, EXTRA = "StringQuality", ("countEmptyString", "countAlphabetic", "countNumerical", "countWordCharacter", "countWhitespace")

    def __init__(self, *a, unique=None, **kw):
        super().__init__(*a, **kw)
        self.unique = set(unique) if unique is not None else set()

    @property
    def countApproxUnique(self):
        return len(self.unique)

    def add(self, other):
        r = super().add(other)
        r.unique = self.unique | other.unique
        return r

    def __repr__(self):
        return super().__repr__()[:-1] + ", countApproxUnique=%d)" % self.countApproxUnique


_RE_ALPHA, _RE_NUM, _RE_WORD, _RE_WS = re.compile(r"[a-zA-Z]\Z"), re.compile(r"\d+\Z"), re.compile(r"\w+\Z"), re.compile(r"\s+\Z")


def quality_of_column(meta: ColumnMetaData, col: Sequence[Writable]) -> ColumnQuality:
    """One pass over a column with the classification rules of analysis/quality/**/*QualityAddFunction.java."""
    t = meta.type
    valid = invalid = missing = 0
    total = len(col)
    if t in (ColumnType.Double, ColumnType.Float):
        nonreal = nan = inf = 0
        for w in col:
            if meta.isValid(w):
                valid += 1
            elif _is_missing(w):
                missing += 1
            else:
                invalid += 1
            try:
                d = float(w.toString())
                nan += math.isnan(d)
                inf += math.isinf(d)
            except (ValueError, TypeError):
                nonreal += 1
        return DoubleQuality(valid, invalid, missing, total, countNonReal=nonreal, countNaN=nan, countInfinite=inf)
    if t in (ColumnType.Integer, ColumnType.Long):
        non = 0
        for w in col:
            if meta.isValid(w):
                valid += 1
            elif _is_missing(w):
                missing += 1
            else:
                invalid += 1
            try:
                int(w.toString())
            except (ValueError, TypeError):
                non += 1
        return IntegerQuality(valid, invalid, missing, total, countNonInteger=non) if t == ColumnType.Integer \
            else LongQuality(valid, invalid, missing, total, countNonLong=non)
    if t == ColumnType.String:
        empty = alpha = num = word = ws = 0
        uniq = set()
        for w in col:
            if isinstance(w, NullWritable) or w is None:
                missing += 1
            elif meta.isValid(w):
                valid += 1
            else:
                invalid += 1
            s = None if w is None else w.toString()
            if s is None or s == "":
                empty += 1
            else:
                alpha += bool(_RE_ALPHA.match(s))
                num += bool(_RE_NUM.match(s))
                word += bool(_RE_WORD.match(s))
                ws += bool(_RE_WS.match(s))
            uniq.add(s)
        return StringQuality(valid, invalid, missing, total, countEmptyString=empty, countAlphabetic=alpha, countNumerical=num,
                             countWordCharacter=word, countWhitespace=ws, unique=uniq)
    cls = {ColumnType.Categorical: CategoricalQuality, ColumnType.Time: TimeQuality, ColumnType.Bytes: BytesQuality}.get(t, ColumnQuality)
    for w in col:
        if _is_missing(w):
            missing += 1
        elif meta.isValid(w):
            valid += 1
        else:
            invalid += 1
    return cls(valid, invalid, missing, total)


class DataQualityAnalysis:
    """DataQualityAnalysis.java: schema + one ColumnQuality per column; ``str()`` is the report table with ok / FAIL per column."""

    def __init__(self, schema: Schema, columnQualityList: List[ColumnQuality]):
        self.schema, self.columnQualityList = schema, list(columnQualityList)

    def getSchema(self): return self.schema
    def getColumnQualityList(self): return self.columnQualityList

    def getColumnQuality(self, name_or_idx) -> ColumnQuality:
        return self.columnQualityList[name_or_idx if isinstance(name_or_idx, int) else self.schema.getIndexOfColumn(name_or_idx)]

    # dict-style access kept for callers of the earlier summary form: q["col"]["valid"]
    def __getitem__(self, name):
        q = self.getColumnQuality(name)
        return {"valid": q.countValid, "invalid": q.countInvalid, "missing": q.countMissing, "total": q.countTotal}

    def __str__(self):
        w = max([len(c.name) for c in self.schema.columns] + [4]) + 8
        out = ["%-6s%-*s%-15s%-10s%s" % ("idx", w, "name", "type", "quality", "details")]
        for i, (c, q) in enumerate(zip(self.schema.columns, self.columnQualityList)):
            ok = q.countInvalid == 0 and q.countMissing == 0
            out.append("%-6d%-*s%-15s%-10s%s" % (i, w, '"%s"' % c.name, c.type, "ok" if ok else "FAIL", q))
        return "\n".join(out) + "\n"


# ================================================================================================ AnalyzeLocal
class AnalyzeLocal:
    """datavec-local AnalyzeLocal.java: analyze / analyzeQuality (+ sequence forms) over a record reader or an iterable of records."""
    DEFAULT_MAX_HISTOGRAM_BUCKETS = 20

    @staticmethod
    def _records(reader_or_records):
        if hasattr(reader_or_records, "hasNext") and hasattr(reader_or_records, "next"):
            out = []
            while reader_or_records.hasNext():
                out.append(reader_or_records.next())
            return out
        return list(reader_or_records)

    @staticmethod
    def analyze(schema: Schema, reader_or_records, maxHistogramBuckets: int = 20) -> DataAnalysis:
        recs = AnalyzeLocal._records(reader_or_records)
        return DataAnalysis(schema, [analyze_column(c, [r[i] for r in recs], maxHistogramBuckets) for i, c in enumerate(schema.columns)])

    @staticmethod
    def analyzeSequence(schema: Schema, sequences, maxHistogramBuckets: int = 20) -> SequenceDataAnalysis:
        seqs = [list(s) for s in (AnalyzeLocal._records(sequences))]
        flat = [step for s in seqs for step in s]
        base = AnalyzeLocal.analyze(schema, flat, maxHistogramBuckets)
        return SequenceDataAnalysis(schema, base.columnAnalysis, SequenceLengthAnalysis([len(s) for s in seqs], maxHistogramBuckets))

    @staticmethod
    def analyzeQuality(schema: Schema, reader_or_records) -> DataQualityAnalysis:
        recs = AnalyzeLocal._records(reader_or_records)
        return DataQualityAnalysis(schema, [quality_of_column(c, [r[i] for r in recs]) for i, c in enumerate(schema.columns)])

    @staticmethod
    def analyzeQualitySequence(schema: Schema, sequences) -> DataQualityAnalysis:
        flat = [step for s in AnalyzeLocal._records(sequences) for step in s]
        return AnalyzeLocal.analyzeQuality(schema, flat)

    @staticmethod
    def getUnique(columnName, schema: Schema, reader_or_records):
        """Unique values of one column (or a dict for a list of columns), first-seen order — AnalyzeLocal.getUnique."""
        recs = AnalyzeLocal._records(reader_or_records)
        names = [columnName] if isinstance(columnName, str) else list(columnName)
        out = {}
        for n in names:
            i = schema.getIndexOfColumn(n)
            seen, vals = set(), []
            for r in recs:
                k = r[i].toString()
                if k not in seen:
                    seen.add(k)
                    vals.append(r[i])
            out[n] = vals
        return out[columnName] if isinstance(columnName, str) else out


# ================================================================================================ HTML report
def _svg_histogram(edges, counts, width=460, height=160, title=""):
    if not edges or not counts:
        return "<div class='nochart'>(no histogram)</div>"
    n = len(counts)
    mx = max(max(counts), 1)
    bw = (width - 50) / n
    bars = []
    for i, c in enumerate(counts):
        h = (height - 40) * c / mx
        bars.append("<rect x='%.1f' y='%.1f' width='%.1f' height='%.1f' fill='#4682b4'><title>[%g, %g): %d</title></rect>" % (
            40 + i * bw, height - 25 - h, max(bw - 1, 1), h, edges[i], edges[i + 1], c))
    axis = ("<line x1='40' y1='%d' x2='%d' y2='%d' stroke='#333'/><text x='40' y='%d' font-size='10'>%g</text>"
            "<text x='%d' y='%d' font-size='10' text-anchor='end'>%g</text><text x='4' y='14' font-size='10'>max %d</text>") % (
        height - 25, width - 10, height - 25, height - 10, edges[0], width - 10, height - 10, edges[-1], mx)
    return "<svg class='hist' width='%d' height='%d' role='img' aria-label='%s'>%s%s</svg>" % (width, height, html.escape(title), "".join(bars), axis)


def _svg_line(xs, series: Dict[str, Sequence[float]], width=640, height=200):
    allv = [v for s in series.values() for v in s if v == v]
    if not xs or not allv:
        return "<div class='nochart'>(no data)</div>"
    lo, hi = min(allv), max(allv)
    span = (hi - lo) or 1.0
    x0, x1 = min(xs), max(xs)
    xs_span = (x1 - x0) or 1.0
    colors = ["#1f77b4", "#d62728", "#2ca02c", "#9467bd", "#ff7f0e", "#8c564b"]
    paths = []
    for k, (name, ys) in enumerate(series.items()):
        pts = " ".join("%.1f,%.1f" % (40 + (width - 60) * (x - x0) / xs_span, height - 25 - (height - 45) * (y - lo) / span)
                       for x, y in zip(xs, ys) if y == y)
        paths.append("<polyline fill='none' stroke='%s' stroke-width='1.5' points='%s'><title>%s</title></polyline>" % (
            colors[k % len(colors)], pts, html.escape(name)))
    axis = "<line x1='40' y1='%d' x2='%d' y2='%d' stroke='#333'/><text x='2' y='%d' font-size='10'>%g</text><text x='2' y='14' font-size='10'>%g</text>" % (
        height - 25, width - 10, height - 25, height - 27, lo, hi)
    return "<svg class='line' width='%d' height='%d'>%s%s</svg>" % (width, height, "".join(paths), axis)


_CSS = ("body{font-family:sans-serif;margin:24px}table{border-collapse:collapse}td,th{border:1px solid #bbb;padding:4px 8px;"
        "font-size:13px;vertical-align:top}th{background:#eee}h2{margin-top:28px}.nochart{color:#777;font-size:12px}")


class HtmlAnalysis:
    """ui/HtmlAnalysis.java: a self-contained HTML page — the summary table of all columns, then one section per column with its
    statistics table and (numeric / string length / NDArray values) histogram. The reference renders through FreeMarker + its
    chart components; this writes inline SVG, no external assets."""

    @staticmethod
    def createHtmlAnalysisString(analysis: _BaseDataAnalysis) -> str:
        schema = analysis.getSchema()
        rows, sections = [], []
        seq = getattr(analysis, "sequenceLengthAnalysis", None)
        for i, (c, a) in enumerate(zip(schema.columns, analysis.columnAnalysis)):
            rows.append("<tr><td>%d</td><td>%s</td><td>%s</td><td>%s</td></tr>" % (i, html.escape(c.name), c.type, html.escape(repr(a))))
            d = a.to_dict() if hasattr(a, "to_dict") else {"countTotal": a.count}
            stat_rows = "".join("<tr><th>%s</th><td>%s</td></tr>" % (html.escape(str(k)), html.escape(str(v)))
                                for k, v in d.items() if k not in ("@class", "histogramBuckets", "histogramBucketCounts", "quantiles"))
            if isinstance(a, CategoricalAnalysis):
                names = list(a.stateCounts)
                chart = _svg_histogram(list(range(len(names) + 1)), [a.stateCounts[n] for n in names], title=c.name)
            else:
                chart = _svg_histogram(getattr(a, "histogramBuckets", None), getattr(a, "histogramCounts", None), title=c.name)
            sections.append("<h2 id='col%d'>%s <small>(%s)</small></h2><table>%s</table>%s" % (i, html.escape(c.name), c.type, stat_rows, chart))
        seq_html = ""
        if seq is not None:
            seq_html = "<h2>Sequence lengths</h2><p>%s</p>%s" % (html.escape(repr(seq)),
                                                               _svg_histogram(seq.histogramBuckets, seq.histogramBucketCounts, title="sequence length"))
        return ("<!DOCTYPE html><html><head><meta charset='utf-8'><title>Data Analysis</title><style>%s</style></head><body>"
                "<h1>Data Analysis</h1><table><tr><th>#</th><th>Column</th><th>Type</th><th>Summary</th></tr>%s</table>%s%s"
                "</body></html>") % (_CSS, "".join(rows), seq_html, "".join(sections))

    @staticmethod
    def createHtmlAnalysisFile(analysis, path):
        with open(path, "w", encoding="utf-8") as f:
            f.write(HtmlAnalysis.createHtmlAnalysisString(analysis))


class HtmlSequencePlotting:
    """ui/HtmlSequencePlotting.java: one line chart per numeric column of a single sequence (categorical columns plot the state
    index), x = time step."""

    @staticmethod
    def createHtmlSequencePlots(title: str, schema: Schema, sequence: Sequence[Sequence[Writable]]) -> str:
        xs = list(range(len(sequence)))
        parts = []
        for i, c in enumerate(schema.columns):
            if c.type in _NUMERIC:
                ys = []
                for step in sequence:
                    try:
                        ys.append(step[i].toDouble())
                    except (ValueError, TypeError):
                        ys.append(float("nan"))
            elif c.type == ColumnType.Categorical:
                names = c.stateNames or []
                ys = [float(names.index(step[i].toString())) if step[i].toString() in names else float("nan") for step in sequence]
            else:
                continue
            parts.append("<h2>%s <small>(%s)</small></h2>%s" % (html.escape(c.name), c.type, _svg_line(xs, {c.name: ys})))
        return ("<!DOCTYPE html><html><head><meta charset='utf-8'><title>%s</title><style>%s</style></head><body><h1>%s</h1>%s"
                "</body></html>") % (html.escape(title), _CSS, html.escape(title), "".join(parts))

    @staticmethod
    def createHtmlSequencePlotFile(title, schema, sequence, path):
        with open(path, "w", encoding="utf-8") as f:
            f.write(HtmlSequencePlotting.createHtmlSequencePlots(title, schema, sequence))


# ================================================================================================ sorted rank
class WritableComparator:
    """org.datavec.api.writable.comparator.*: compare(a, b) -> negative / 0 / positive."""
    NAME = "WritableComparator"

    def key(self, w):
        raise NotImplementedError

    def compare(self, a: Writable, b: Writable) -> int:
        ka, kb = self.key(a), self.key(b)
        return (ka > kb) - (ka < kb)

    def to_json(self):
        return {"@class": self.NAME}

    def __eq__(self, o):
        return type(o) is type(self)

    def __repr__(self):
        return self.NAME + "()"


class DoubleWritableComparator(WritableComparator):
    NAME = "DoubleWritableComparator"
    def key(self, w): return w.toDouble()


class FloatWritableComparator(DoubleWritableComparator):
    NAME = "FloatWritableComparator"


class IntWritableComparator(WritableComparator):
    NAME = "IntWritableComparator"
    def key(self, w): return int(w.toDouble())


class LongWritableComparator(IntWritableComparator):
    NAME = "LongWritableComparator"


class TextWritableComparator(WritableComparator):
    NAME = "TextWritableComparator"
    def key(self, w): return w.toString()


class ReverseComparator(WritableComparator):
    NAME = "ReverseComparator"

    def __init__(self, inner: WritableComparator):
        self.inner = inner

    def compare(self, a, b):
        return -self.inner.compare(a, b)

    def to_json(self):
        return {"@class": self.NAME, "comparator": self.inner.to_json()}

    def __eq__(self, o):
        return isinstance(o, ReverseComparator) and o.inner == self.inner


_COMPARATORS = {c.NAME: c for c in (DoubleWritableComparator, FloatWritableComparator, IntWritableComparator, LongWritableComparator,
                                    TextWritableComparator)}


def comparator_from_json(d) -> WritableComparator:
    if d["@class"].endswith("ReverseComparator"):
        return ReverseComparator(comparator_from_json(d["comparator"]))
    return _COMPARATORS[d["@class"].rsplit(".", 1)[-1]]()


def default_comparator(columnType: str) -> WritableComparator:
    if columnType in (ColumnType.Integer,):
        return IntWritableComparator()
    if columnType in (ColumnType.Long, ColumnType.Time):
        return LongWritableComparator()
    if columnType in (ColumnType.Double, ColumnType.Float):
        return DoubleWritableComparator()
    return TextWritableComparator()


class CalculateSortedRank(_SortedRank):
    """rank/CalculateSortedRank.java: sort all examples on one column with a WritableComparator and append the rank (Long, 0-based)
    as a new column. As in LocalTransformExecutor.java:469-500 the OUTPUT ORDER is the sorted order. Not defined on sequences."""

    def __init__(self, newColumnName: str, sortOnColumn: str, comparator: Optional[WritableComparator] = None, ascending: bool = True):
        _SortedRank.__init__(self, newColumnName, sortOnColumn, bool(ascending))      # newCol / sortCol / ascending of the step machinery
        self.newColumnName, self.sortOnColumn, self.comparator = newColumnName, sortOnColumn, comparator
        self.inputSchema: Optional[Schema] = None

    def transform(self, inputSchema: Schema) -> Schema:
        if getattr(inputSchema, "isSequence", False):
            raise ValueError("Calculating sorted rank on sequences: not yet supported")
        inputSchema.getIndexOfColumn(self.sortOnColumn)
        return Schema(list(inputSchema.columns) + [ColumnMetaData(self.newColumnName, ColumnType.Long, None, 0, None)])

    def setInputSchema(self, schema: Schema):
        self.inputSchema = schema
        if self.comparator is None:
            self.comparator = default_comparator(schema.columns[schema.getIndexOfColumn(self.sortOnColumn)].type)

    def getInputSchema(self): return self.inputSchema
    def outputColumnName(self): return self.newColumnName
    def outputColumnNames(self): return [c.name for c in self.inputSchema.columns] + [self.newColumnName]
    def columnNames(self): return [c.name for c in self.inputSchema.columns]
    def columnName(self): return self.columnNames()[0]

    def execute(self, records: List[List[Writable]], schema: Optional[Schema] = None) -> List[List[Writable]]:
        schema = schema or self.inputSchema
        if self.comparator is None:
            self.setInputSchema(schema)
        i = schema.getIndexOfColumn(self.sortOnColumn)
        import functools
        sign = 1 if self.ascending else -1
        ordered = sorted(records, key=functools.cmp_to_key(lambda a, b: sign * self.comparator.compare(a[i], b[i])))
        return [list(r) + [LongWritable(rank)] for rank, r in enumerate(ordered)]

    def to_json(self):
        return {"@class": "org.datavec.api.transform.rank.CalculateSortedRank", "newColumnName": self.newColumnName,
                "sortOnColumn": self.sortOnColumn, "comparator": None if self.comparator is None else self.comparator.to_json(),
                "ascending": self.ascending}

    @staticmethod
    def from_json(d):
        c = d.get("comparator")
        return CalculateSortedRank(d["newColumnName"], d["sortOnColumn"], comparator_from_json(c) if c else None, d.get("ascending", True))

    def __eq__(self, o):
        return isinstance(o, CalculateSortedRank) and self.to_json() == o.to_json()

    def __repr__(self):
        return 'CalculateSortedRank(newColumnName="%s", comparator=%s)' % (self.newColumnName, self.comparator)
