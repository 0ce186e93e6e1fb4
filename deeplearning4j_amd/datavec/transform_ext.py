"""DataVec transforms, conditions, filters, sequence operations and utilities, second tranche (reference: datavec-api/src/main/java/
org/datavec/api/transform/**: transform/categorical/{FirstDigitTransform,PivotTransform}.java, transform/integer/ReplaceEmptyIntegerWithValue
Transform.java, transform/nlp/{TextToCharacterIndexTransform,TextToTermIndexSequenceTransform}.java, transform/parse/ParseDoubleTransform.java,
transform/sequence/SequenceDifferenceTransform.java, transform/string/{MapAllStringsExceptListTransform,StringListToCategoricalSetTransform,
StringListToCountsNDArrayTransform,StringListToIndicesNDArrayTransform}.java, transform/time/DeriveColumnsFromTimeTransform.java,
transform/doubletransform/Log2Normalizer.java, ndarray/*.java, condition/column/{Boolean,Float,Long,Time,Infinite,Trivial}ColumnCondition.java,
condition/sequence/SequenceLengthCondition.java, condition/string/StringRegexColumnCondition.java, filter/InvalidNumColumns.java,
sequence/{ReduceSequenceTransform,merge/SequenceMerge,split/SequenceSplitTimeSeparation,trim/SequenceTrimToLengthTransform,
window/{TimeWindowFunction,OverlappingTimeWindowFunction,ReduceSequenceByWindowTransform},expansion/BaseSequenceExpansionTransform}.java,
split/RandomSplit.java, stringreduce/StringReducer.java).

Everything builds on the step protocol of ``transform.py`` (bind / outputSchema / map / mapSequence); steps that change the number of
sequences implement ``runAll(data, schema, is_seq)`` which the local executor calls with the whole data set. Importing this module adds
the matching builder methods to ``TransformProcess.Builder``.
"""
from __future__ import annotations

import datetime as _dt
import math
import random
import re
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import transform as T
from .transform import (ColumnCondition, ColumnMetaData, ColumnType, Condition, ConditionOp, MathOp, ReduceOp, Reducer, Schema, TransformProcess,
                        _ColumnMap, _Step, _cmp, _math, _reduce)
from .writable import (BooleanWritable, DoubleWritable, IntWritable, LongWritable, NDArrayWritable, NullWritable, Text, Writable, wrap)


# ----------------------------------------------------------------------------- conditions
class BooleanColumnCondition(Condition):
    """True where the (boolean) column is true."""

    def __init__(self, column):
        self.column = column

    def bind(self, schema):
        self._idx = schema.getIndexOfColumn(self.column)
        return self

    def condition(self, record):
        w = record[self._idx]
        return bool(w.value) if isinstance(w, BooleanWritable) else w.toString().strip().lower() in ("true", "1")


class FloatColumnCondition(T.DoubleColumnCondition):
    pass


class LongColumnCondition(T.IntegerColumnCondition):
    pass


class TimeColumnCondition(T.IntegerColumnCondition):
    pass


class InfiniteColumnCondition(T.NullWritableColumnCondition):
    def condition(self, record):
        try:
            return math.isinf(record[self._idx].toDouble())
        except (ValueError, TypeError):
            return False


class TrivialColumnCondition(T.NullWritableColumnCondition):
    """Always true (used to apply a conditional transform to every row)."""

    def condition(self, record):
        return True


class StringRegexColumnCondition(Condition):
    """True where the whole string matches the regular expression (String.matches semantics)."""

    def __init__(self, column, regex):
        self.column, self.regex = column, regex
        self._re = re.compile(regex)

    def bind(self, schema):
        self._idx = schema.getIndexOfColumn(self.column)
        return self

    def condition(self, record):
        return self._re.fullmatch(record[self._idx].toString()) is not None


class SequenceLengthCondition(Condition):
    """Condition on the number of steps of a sequence: (op, length) or an inclusive [minLength, maxLength] range."""

    def __init__(self, op=None, length=None, minLength=None, maxLength=None):
        self.op, self.length, self.minLength, self.maxLength = op, length, minLength, maxLength

    def bind(self, schema):
        return self

    def condition(self, record):
        raise ValueError("SequenceLengthCondition applies to sequences only")

    def conditionSequence(self, seq):
        n = len(seq)
        if self.op is not None:
            return _cmp(self.op, n, self.length)
        return (self.minLength is None or n >= self.minLength) and (self.maxLength is None or n <= self.maxLength)


# ----------------------------------------------------------------------------- filters
class InvalidNumColumns(_Step):
    """Drops records (or sequences containing records) whose column count differs from the schema's."""

    def map(self, r):
        return r if len(r) == self.inputSchema.numColumns() else None

    def mapSequence(self, seq):
        return None if any(len(r) != self.inputSchema.numColumns() for r in seq) else seq


# ----------------------------------------------------------------------------- categorical / integer / parse
class FirstDigitTransform(_Step):
    """Benford-style first significant digit of a numeric column as a categorical column; invalid / zero values either map to the
    ``Other`` state (mode IncludeOtherCategory) or raise (ExceptionOnInvalid)."""
    ExceptionOnInvalid, IncludeOtherCategory = "EXCEPTION_ON_INVALID", "INCLUDE_OTHER_CATEGORY"
    OTHER = "Other"

    def __init__(self, inputColumn, outputColumn, mode=IncludeOtherCategory):
        self.inputColumn, self.outputColumn, self.mode = inputColumn, outputColumn, mode

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.inputColumn)
        return super().bind(s)

    def _states(self):
        return [str(d) for d in range(10)] + ([self.OTHER] if self.mode == self.IncludeOtherCategory else [])

    def outputSchema(self, s):
        meta = ColumnMetaData(self.outputColumn, ColumnType.Categorical, self._states())
        cols = list(s.columns)
        if self.outputColumn == self.inputColumn:
            cols[s.getIndexOfColumn(self.inputColumn)] = meta
        else:
            cols.insert(s.getIndexOfColumn(self.inputColumn) + 1, meta)
        return Schema(cols)

    def _digit(self, w):
        s = w.toString().strip()
        try:
            v = float(s)
        except ValueError:
            v = float("nan")
        if math.isnan(v) or math.isinf(v):
            if self.mode == self.ExceptionOnInvalid:
                raise ValueError("FirstDigitTransform: invalid numeric value %r in column %s" % (s, self.inputColumn))
            return self.OTHER
        return "0" if v == 0 else ("%.15e" % abs(v))[0]

    def map(self, r):
        r = list(r)
        d = Text(self._digit(r[self._idx]))
        if self.outputColumn == self.inputColumn:
            r[self._idx] = d
        else:
            r.insert(self._idx + 1, d)
        return r


class PivotTransform(_Step):
    """Spreads a value column over one new column per state of a categorical key column: the column of the record's key gets the value,
    the others the default (PivotTransform.java; new columns are named ``key[state]``)."""

    def __init__(self, keyColumn, valueColumn, defaultValue=None):
        self.keyColumn, self.valueColumn, self.defaultValue = keyColumn, valueColumn, defaultValue

    def bind(self, s):
        self._k, self._v = s.getIndexOfColumn(self.keyColumn), s.getIndexOfColumn(self.valueColumn)
        self._states = s.getMetaData(self.keyColumn).stateNames
        if not self._states:
            raise ValueError("PivotTransform: key column %s must be categorical" % self.keyColumn)
        vt = s.getMetaData(self.valueColumn).type
        self._default = wrap(self.defaultValue) if self.defaultValue is not None else (
            Text("") if vt in (ColumnType.String, ColumnType.Categorical) else (DoubleWritable(0.0) if vt in (ColumnType.Double, ColumnType.Float) else IntWritable(0)))
        return super().bind(s)

    def outputSchema(self, s):
        vm = s.getMetaData(self.valueColumn)
        states = s.getMetaData(self.keyColumn).stateNames
        cols = []
        for c in s.columns:
            if c.name == self.keyColumn:
                cols += [ColumnMetaData("%s[%s]" % (self.keyColumn, st), vm.type, vm.stateNames) for st in states]
            elif c.name != self.valueColumn:
                cols.append(c)
        return Schema(cols)

    def map(self, r):
        key = r[self._k].toString()
        out = []
        for i, w in enumerate(r):
            if i == self._k:
                out += [r[self._v] if st == key else self._default for st in self._states]
            elif i != self._v:
                out.append(w)
        return out


class ReplaceEmptyIntegerWithValue(_ColumnMap):
    newType = ColumnType.Integer

    def __init__(self, column, value):
        super().__init__(column)
        self.value = int(value)

    def fn(self, w):
        s = w.toString()
        return IntWritable(self.value) if (isinstance(w, NullWritable) or s is None or s.strip() == "") else w


class ParseDoubleTransform(_Step):
    """Every String column that parses becomes a Double column."""

    def outputSchema(self, s):
        return Schema([ColumnMetaData(c.name, ColumnType.Double) if c.type == ColumnType.String else c for c in s.columns])

    def bind(self, s):
        self._idx = [i for i, c in enumerate(s.columns) if c.type == ColumnType.String]
        return super().bind(s)

    def map(self, r):
        r = list(r)
        for i in self._idx:
            r[i] = DoubleWritable(float(r[i].toString()))
        return r


class Log2Normalizer(_ColumnMap):
    """y = scale * log2((x - min) / (mean - min) + 1) (Log2Normalizer.java)."""
    newType = ColumnType.Double

    def __init__(self, column, columnMean, columnMin, scalingFactor):
        super().__init__(column)
        if columnMean <= columnMin:
            raise ValueError("Log2Normalizer: mean must be greater than min")
        self.columnMean, self.columnMin, self.scalingFactor = float(columnMean), float(columnMin), float(scalingFactor)

    def fn(self, w):
        v = max((w.toDouble() - self.columnMin) / (self.columnMean - self.columnMin) + 1.0, 1.0)
        return DoubleWritable(self.scalingFactor * math.log2(v))


# ----------------------------------------------------------------------------- strings
class MapAllStringsExceptList(_ColumnMap):
    def __init__(self, column, newValue, exceptions):
        super().__init__(column)
        self.newValue, self.exceptions = newValue, set(exceptions)

    def fn(self, w):
        return w if w.toString() in self.exceptions else Text(self.newValue)


class StringListToCategoricalSet(_Step):
    """A delimited list of tokens -> one boolean-like categorical column ("true"/"false") per known token."""

    def __init__(self, column, newColumnNames, categoryTokens, delimiter=","):
        if len(newColumnNames) != len(categoryTokens):
            raise ValueError("StringListToCategoricalSet: one new column name per token is required")
        self.column, self.newColumnNames, self.categoryTokens, self.delimiter = column, list(newColumnNames), list(categoryTokens), delimiter

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.column)
        return super().bind(s)

    def outputSchema(self, s):
        cols = []
        for c in s.columns:
            if c.name == self.column:
                cols += [ColumnMetaData(n, ColumnType.Categorical, ["true", "false"]) for n in self.newColumnNames]
            else:
                cols.append(c)
        return Schema(cols)

    def map(self, r):
        toks = {t.strip() for t in r[self._idx].toString().split(self.delimiter)} if r[self._idx].toString() else set()
        unknown = toks - set(self.categoryTokens) - {""}
        if unknown:
            raise ValueError("StringListToCategoricalSet: unknown token(s) %s in column %s" % (sorted(unknown), self.column))
        return list(r[:self._idx]) + [Text("true" if t in toks else "false") for t in self.categoryTokens] + list(r[self._idx + 1:])


class StringListToCountsNDArray(_Step):
    """Delimited token list -> NDArray column of per-vocabulary-entry counts (or 0/1 with ``binary``)."""

    def __init__(self, column, vocabulary, delimiter=",", newColumnName=None, binary=False, ignoreUnknown=False):
        self.column, self.vocabulary, self.delimiter = column, list(vocabulary), delimiter
        self.newColumnName, self.binary, self.ignoreUnknown = newColumnName or column, binary, ignoreUnknown
        self._map = {t: i for i, t in enumerate(self.vocabulary)}

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.column)
        return super().bind(s)

    def outputSchema(self, s):
        cols = list(s.columns)
        cols[s.getIndexOfColumn(self.column)] = ColumnMetaData(self.newColumnName, ColumnType.NDArray)
        return Schema(cols)

    def _indices(self, text):
        out = []
        for t in (text.split(self.delimiter) if text else []):
            t = t.strip()
            if t in self._map:
                out.append(self._map[t])
            elif not self.ignoreUnknown and t:
                raise ValueError("unknown token %r in column %s" % (t, self.column))
        return out

    def _array(self, idx):
        a = np.zeros((1, len(self.vocabulary)), dtype=np.float32)
        for i in idx:
            a[0, i] = 1.0 if self.binary else a[0, i] + 1.0
        return a

    def map(self, r):
        r = list(r)
        r[self._idx] = NDArrayWritable(self._array(self._indices(r[self._idx].toString())))
        return r


class StringListToIndicesNDArray(StringListToCountsNDArray):
    """Delimited token list -> NDArray column holding the vocabulary indices of the tokens in order."""

    def _array(self, idx):
        return np.asarray(idx, dtype=np.float32).reshape(1, -1)


# ----------------------------------------------------------------------------- NLP
class TextToCharacterIndexTransform(_Step):
    """Sequence expansion: every record of the sequence becomes one record per character of the text column, holding the character's
    index in ``characterIndexMap`` (BaseSequenceExpansionTransform with one expanded column; the other columns are copied)."""

    def __init__(self, column, newColumnName, characterIndexMap: Dict[str, int], exceptionOnUnknown=True):
        self.column, self.newColumnName, self.characterIndexMap, self.exceptionOnUnknown = column, newColumnName, dict(characterIndexMap), exceptionOnUnknown

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.column)
        return super().bind(s)

    def outputSchema(self, s):
        cols = list(s.columns)
        hi = max(self.characterIndexMap.values()) if self.characterIndexMap else 0
        cols[s.getIndexOfColumn(self.column)] = ColumnMetaData(self.newColumnName, ColumnType.Integer, minAllowed=0, maxAllowed=hi)
        return Schema(cols)

    def expand(self, w) -> List[Writable]:
        out = []
        for ch in w.toString():
            if ch in self.characterIndexMap:
                out.append(IntWritable(self.characterIndexMap[ch]))
            elif self.exceptionOnUnknown:
                raise ValueError("TextToCharacterIndexTransform: unknown character %r" % ch)
        return out

    def map(self, r):
        raise ValueError("%s expands sequences; convert to a sequence first" % type(self).__name__)

    def mapSequence(self, seq):
        out = []
        for r in seq:
            for v in self.expand(r[self._idx]):
                row = list(r)
                row[self._idx] = v
                out.append(row)
        return out


BaseSequenceExpansionTransform = TextToCharacterIndexTransform


class TextToTermIndexSequenceTransform(TextToCharacterIndexTransform):
    """As above with delimiter-separated terms and a word -> index map."""

    def __init__(self, column, newColumnName, wordIndexMap: Dict[str, int], delimiter=" ", exceptionOnUnknown=True):
        super().__init__(column, newColumnName, wordIndexMap, exceptionOnUnknown)
        self.delimiter = delimiter

    def expand(self, w):
        out = []
        for t in w.toString().split(self.delimiter):
            if t in self.characterIndexMap:
                out.append(IntWritable(self.characterIndexMap[t]))
            elif self.exceptionOnUnknown and t:
                raise ValueError("TextToTermIndexSequenceTransform: unknown term %r" % t)
        return out


# ----------------------------------------------------------------------------- time
class DeriveColumnsFromTime(_Step):
    """New columns derived from a Time column (epoch milliseconds, UTC or a fixed offset in hours): calendar fields as Integer columns
    (``addIntegerDerivedColumn``) or a formatted String column (``addStringDerivedColumn``, strftime pattern); inserted after ``insertAfter``."""
    FIELDS = ("year", "monthOfYear", "dayOfMonth", "dayOfWeek", "dayOfYear", "hourOfDay", "minuteOfHour", "secondOfMinute", "millisOfSecond",
              "minuteOfDay", "secondOfDay", "weekOfWeekyear")

    def __init__(self, timeColumn, derived: Sequence[tuple], insertAfter=None, utcOffsetHours=0.0):
        self.timeColumn, self.derived, self.insertAfter, self.utcOffsetHours = timeColumn, list(derived), insertAfter or timeColumn, float(utcOffsetHours)
        for name, kind, arg in self.derived:
            if kind == "Integer" and arg not in self.FIELDS:
                raise ValueError("DeriveColumnsFromTime: unknown field %r (have %s)" % (arg, ", ".join(self.FIELDS)))

    class Builder:
        def __init__(self, timeColumn):
            self._c, self._d, self._after, self._off = timeColumn, [], None, 0.0

        def insertAfter(self, col):
            self._after = col
            return self

        def timeZoneOffsetHours(self, h):
            self._off = h
            return self

        def addIntegerDerivedColumn(self, name, field):
            self._d.append((name, "Integer", field))
            return self

        def addStringDerivedColumn(self, name, fmt, tz=None):
            self._d.append((name, "String", fmt))
            return self

        def build(self):
            return DeriveColumnsFromTime(self._c, self._d, self._after, self._off)

    def bind(self, s):
        self._t, self._pos = s.getIndexOfColumn(self.timeColumn), s.getIndexOfColumn(self.insertAfter) + 1
        return super().bind(s)

    def outputSchema(self, s):
        cols = list(s.columns)
        pos = s.getIndexOfColumn(self.insertAfter) + 1
        new = [ColumnMetaData(n, ColumnType.Integer if k == "Integer" else ColumnType.String) for n, k, _ in self.derived]
        return Schema(cols[:pos] + new + cols[pos:])

    def _value(self, ms, kind, arg):
        d = _dt.datetime(1970, 1, 1) + _dt.timedelta(milliseconds=int(ms)) + _dt.timedelta(hours=self.utcOffsetHours)
        if kind == "String":
            return Text(d.strftime(arg))
        v = {"year": d.year, "monthOfYear": d.month, "dayOfMonth": d.day, "dayOfWeek": d.isoweekday(), "dayOfYear": d.timetuple().tm_yday,
             "hourOfDay": d.hour, "minuteOfHour": d.minute, "secondOfMinute": d.second, "millisOfSecond": d.microsecond // 1000,
             "minuteOfDay": d.hour * 60 + d.minute, "secondOfDay": d.hour * 3600 + d.minute * 60 + d.second, "weekOfWeekyear": d.isocalendar()[1]}[arg]
        return IntWritable(v)

    def map(self, r):
        ms = int(r[self._t].toDouble())
        new = [self._value(ms, k, a) for _, k, a in self.derived]
        return list(r[:self._pos]) + new + list(r[self._pos:])


# ----------------------------------------------------------------------------- NDArray columns
def _arr(w):
    import torch
    return torch.as_tensor(w.value if isinstance(w, NDArrayWritable) else w.toDouble()).double()


class NDArrayScalarOpTransform(_ColumnMap):
    def __init__(self, column, op, scalar):
        super().__init__(column)
        self.op, self.scalar = op, float(scalar)

    def fn(self, w):
        a = _arr(w)
        s = self.scalar
        out = {MathOp.Add: a + s, MathOp.Subtract: a - s, MathOp.Multiply: a * s, MathOp.Divide: a / s, MathOp.Modulus: a % s,
               MathOp.ReverseSubtract: s - a, MathOp.ReverseDivide: s / a, MathOp.ScalarMin: a.clamp(max=s), MathOp.ScalarMax: a.clamp(min=s)}[self.op]
        return NDArrayWritable(out.to(getattr(w.value, "dtype", out.dtype)) if hasattr(w.value, "dtype") and not isinstance(w.value, np.ndarray) else out.float())


class NDArrayMathFunctionTransform(_ColumnMap):
    def __init__(self, column, function):
        super().__init__(column)
        self.function = function

    def fn(self, w):
        import torch
        a = _arr(w)
        f = {"ABS": torch.abs, "ACOS": torch.acos, "ASIN": torch.asin, "ATAN": torch.atan, "CEIL": torch.ceil, "COS": torch.cos, "COSH": torch.cosh,
             "EXP": torch.exp, "FLOOR": torch.floor, "LOG": torch.log, "LOG10": torch.log10, "SIGNUM": torch.sign, "SIN": torch.sin, "SINH": torch.sinh,
             "SQRT": torch.sqrt, "TAN": torch.tan, "TANH": torch.tanh}[str(self.function).upper()]
        return NDArrayWritable(f(a).float())


class NDArrayColumnsMathOpTransform(_Step):
    """Element-wise Add / Subtract / Multiply / Divide / Modulus over several NDArray columns of equal shape -> new column."""

    def __init__(self, newColumnName, op, *columns):
        self.newColumnName, self.op, self.columns = newColumnName, op, list(columns)

    def bind(self, s):
        self._idx = [s.getIndexOfColumn(c) for c in self.columns]
        return super().bind(s)

    def outputSchema(self, s):
        return Schema(list(s.columns) + [ColumnMetaData(self.newColumnName, ColumnType.NDArray)])

    def map(self, r):
        arrs = [_arr(r[i]) for i in self._idx]
        out = arrs[0].clone()
        for a in arrs[1:]:
            if a.shape != out.shape:
                raise ValueError("NDArrayColumnsMathOpTransform: shapes differ: %s vs %s" % (tuple(out.shape), tuple(a.shape)))
            out = {MathOp.Add: out + a, MathOp.Subtract: out - a, MathOp.Multiply: out * a, MathOp.Divide: out / a, MathOp.Modulus: out % a}[self.op]
        return list(r) + [NDArrayWritable(out.float())]


class Distance:
    COSINE, EUCLIDEAN, MANHATTAN = "COSINE", "EUCLIDEAN", "MANHATTAN"


class NDArrayDistanceTransform(_Step):
    def __init__(self, newColumnName, distance, firstCol, secondCol):
        self.newColumnName, self.distance, self.firstCol, self.secondCol = newColumnName, distance, firstCol, secondCol

    def bind(self, s):
        self._a, self._b = s.getIndexOfColumn(self.firstCol), s.getIndexOfColumn(self.secondCol)
        return super().bind(s)

    def outputSchema(self, s):
        return Schema(list(s.columns) + [ColumnMetaData(self.newColumnName, ColumnType.Double)])

    def map(self, r):
        a, b = _arr(r[self._a]).reshape(-1), _arr(r[self._b]).reshape(-1)
        if self.distance == Distance.COSINE:
            d = 1.0 - float((a @ b) / (a.norm() * b.norm()))
        elif self.distance == Distance.EUCLIDEAN:
            d = float((a - b).norm())
        elif self.distance == Distance.MANHATTAN:
            d = float((a - b).abs().sum())
        else:
            raise ValueError("unknown distance %r" % self.distance)
        return list(r) + [DoubleWritable(d)]


# ----------------------------------------------------------------------------- sequence operations
class SequenceDifferenceTransform(_Step):
    """x[t] - x[t - lookback]; the first ``lookback`` steps get the default (FirstStepMode.Default, 0) or a given value (SpecifiedValue)."""
    Default, SpecifiedValue = "Default", "SpecifiedValue"

    def __init__(self, column, newColumnName=None, lookback=1, firstStepMode=Default, specifiedValue=None):
        self.column, self.newColumnName, self.lookback = column, newColumnName or column, int(lookback)
        self.firstStepMode, self.specifiedValue = firstStepMode, specifiedValue

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.column)
        self._int = s.getMetaData(self.column).type in (ColumnType.Integer, ColumnType.Long, ColumnType.Time)
        return super().bind(s)

    def outputSchema(self, s):
        cols = list(s.columns)
        i = s.getIndexOfColumn(self.column)
        cols[i] = ColumnMetaData(self.newColumnName, cols[i].type)
        return Schema(cols)

    def map(self, r):
        raise ValueError("SequenceDifferenceTransform applies to sequences only")

    def mapSequence(self, seq):
        out = []
        for t, r in enumerate(seq):
            row = list(r)
            if t < self.lookback:
                v = wrap(self.specifiedValue) if self.firstStepMode == self.SpecifiedValue else (LongWritable(0) if self._int else DoubleWritable(0.0))
            else:
                d = r[self._idx].toDouble() - seq[t - self.lookback][self._idx].toDouble()
                v = LongWritable(int(d)) if self._int else DoubleWritable(d)
            row[self._idx] = v
            out.append(row)
        return out


class SequenceTrimToLength(_Step):
    """Trim to ``maxLength``; with mode TrimOrPad shorter sequences are padded with the given record."""
    Trim, TrimOrPad = "TRIM", "TRIM_OR_PAD"

    def __init__(self, maxLength, mode=Trim, pad: Optional[List] = None):
        if mode == self.TrimOrPad and pad is None:
            raise ValueError("SequenceTrimToLength: TRIM_OR_PAD needs a padding record")
        self.maxLength, self.mode, self.pad = int(maxLength), mode, ([wrap(v) for v in pad] if pad is not None else None)

    def map(self, r):
        raise ValueError("SequenceTrimToLength applies to sequences only")

    def mapSequence(self, seq):
        seq = list(seq[:self.maxLength])
        if self.mode == self.TrimOrPad:
            seq += [list(self.pad) for _ in range(self.maxLength - len(seq))]
        return seq


class ReduceSequenceTransform(_Step):
    """Reduce every sequence to a single record with a Reducer (sequence of length 1 afterwards, schema = the reducer's output)."""

    def __init__(self, reducer: Reducer):
        self.reducer = reducer

    def outputSchema(self, s):
        return self.reducer.outputSchema(s)

    def map(self, r):
        raise ValueError("ReduceSequenceTransform applies to sequences only")

    def mapSequence(self, seq):
        saved = self.reducer.keys
        try:
            self.reducer.keys = []                      # the whole sequence is one group
            out = self.reducer.reduce(self.inputSchema, seq)
        finally:
            self.reducer.keys = saved
        return out


class SequenceSplitTimeSeparation(_Step):
    """Split a sequence wherever two consecutive steps are further apart in ``timeColumn`` than the given gap."""
    _UNITS = {"MILLISECONDS": 1, "SECONDS": 1000, "MINUTES": 60000, "HOURS": 3600000, "DAYS": 86400000}

    def __init__(self, timeColumn, timeQuantity, timeUnit="MILLISECONDS"):
        self.timeColumn, self.timeQuantity, self.timeUnit = timeColumn, timeQuantity, timeUnit
        self._gap = int(timeQuantity) * self._UNITS[timeUnit]

    def bind(self, s):
        self._idx = s.getIndexOfColumn(self.timeColumn)
        return super().bind(s)

    def split(self, seq):
        out, cur, last = [], [], None
        for r in seq:
            t = int(r[self._idx].toDouble())
            if last is not None and t - last > self._gap and cur:
                out.append(cur)
                cur = []
            cur.append(r)
            last = t
        if cur:
            out.append(cur)
        return out

    def runAll(self, data, schema, is_seq):
        if not is_seq:
            raise ValueError("SequenceSplitTimeSeparation applies to sequences only")
        return [p for seq in data for p in self.split(seq)], True


class TimeWindowFunction:
    """Non-overlapping windows of fixed duration aligned to ``offset``; empty windows are dropped unless ``addWindowStartTimeColumn`` etc.
    are requested (WindowFunction.applyToSequence)."""

    def __init__(self, timeColumn, windowSize, windowSizeUnit="MILLISECONDS", offset=0, addWindowStartTimeColumn=False, addWindowEndTimeColumn=False,
                 excludeEmptyWindows=True):
        self.timeColumn = timeColumn
        self.size = int(windowSize) * SequenceSplitTimeSeparation._UNITS[windowSizeUnit]
        self.offset, self.addStart, self.addEnd, self.excludeEmpty = int(offset), addWindowStartTimeColumn, addWindowEndTimeColumn, excludeEmptyWindows
        self.step = self.size

    def outputSchema(self, s):
        cols = list(s.columns)
        if self.addStart:
            cols.append(ColumnMetaData("windowStartTime", ColumnType.Time))
        if self.addEnd:
            cols.append(ColumnMetaData("windowEndTime", ColumnType.Time))
        return Schema(cols)

    def applyToSequence(self, schema, seq):
        if not seq:
            return []
        i = schema.getIndexOfColumn(self.timeColumn)
        times = [int(r[i].toDouble()) for r in seq]
        first = self.offset + ((min(times) - self.offset) // self.step) * self.step - (self.size - self.step)
        out, start = [], first
        while start <= max(times):
            end = start + self.size
            win = [list(r) + ([LongWritable(start)] if self.addStart else []) + ([LongWritable(end)] if self.addEnd else [])
                   for r, t in zip(seq, times) if start <= t < end]
            if win or not self.excludeEmpty:
                out.append(win)
            start += self.step
        return out


class OverlappingTimeWindowFunction(TimeWindowFunction):
    """Windows of ``windowSize`` starting every ``windowSeparation``."""

    def __init__(self, timeColumn, windowSize, windowSizeUnit, windowSeparation, windowSeparationUnit="MILLISECONDS", offset=0,
                 addWindowStartTimeColumn=False, addWindowEndTimeColumn=False, excludeEmptyWindows=True):
        super().__init__(timeColumn, windowSize, windowSizeUnit, offset, addWindowStartTimeColumn, addWindowEndTimeColumn, excludeEmptyWindows)
        self.step = int(windowSeparation) * SequenceSplitTimeSeparation._UNITS[windowSeparationUnit]
        if self.step <= 0 or self.step > self.size:
            raise ValueError("OverlappingTimeWindowFunction: separation must be in (0, window size]")


class ReduceSequenceByWindowTransform(_Step):
    """Window every sequence, reduce each window to one record: sequence in, (shorter) sequence out."""

    def __init__(self, reducer: Reducer, windowFunction: TimeWindowFunction):
        self.reducer, self.windowFunction = reducer, windowFunction

    def outputSchema(self, s):
        return self.reducer.outputSchema(self.windowFunction.outputSchema(s))

    def map(self, r):
        raise ValueError("ReduceSequenceByWindowTransform applies to sequences only")

    def mapSequence(self, seq):
        ws = self.windowFunction.outputSchema(self.inputSchema)
        red = ReduceSequenceTransform(self.reducer).bind(ws)
        out = []
        for win in self.windowFunction.applyToSequence(self.inputSchema, seq):
            if win:
                out += red.mapSequence(win)
        return out


class SequenceMerge:
    """Merge several sequences (same schema) into one, ordered by a comparator (sequence/merge/SequenceMerge.java)."""

    def __init__(self, comparator):
        self.comparator = comparator

    def mergeSequences(self, schema: Schema, sequences):
        i = schema.getIndexOfColumn(self.comparator.column)
        num = not isinstance(self.comparator, T.StringComparator)
        rows = [r for seq in sequences for r in seq]
        rows.sort(key=(lambda r: r[i].toDouble()) if num else (lambda r: r[i].toString()), reverse=not self.comparator.ascending)
        return rows


# ----------------------------------------------------------------------------- split / string reduce
class RandomSplit:
    """SplitStrategy: each example goes to the training side with probability ``fractionTrain``."""

    def __init__(self, fractionTrain, seed=None):
        if not 0.0 <= fractionTrain <= 1.0:
            raise ValueError("RandomSplit: fractionTrain must be in [0, 1]")
        self.fractionTrain, self.seed = float(fractionTrain), seed

    def split(self, records):
        rng = random.Random(self.seed)
        train, test = [], []
        for r in records:
            (train if rng.random() < self.fractionTrain else test).append(r)
        return train, test


class StringReduceOp:
    PREPEND, APPEND, MERGE, REPLACE = "PREPEND", "APPEND", "MERGE", "REPLACE"


class StringReducer(_Step):
    """Combine several string columns of a record into one output column (stringreduce/StringReducer.java): APPEND / MERGE concatenate
    (MERGE with a delimiter), PREPEND concatenates in reverse order, REPLACE takes the last column's value."""

    def __init__(self, inputColumns, op, outputColumnName, delimiter=""):
        self.inputColumns, self.op, self.outputColumnName, self.delimiter = list(inputColumns), op, outputColumnName, delimiter

    class Builder:
        def __init__(self, op):
            self._op, self._cols, self._out, self._delim = op, [], None, ""

        def appendColumns(self, *c):
            self._cols += list(c)
            return self

        prependColumns = mergeColumns = replaceColumn = appendColumns

        def outputColumnName(self, n):
            self._out = n
            return self

        def delimiter(self, d):
            self._delim = d
            return self

        def build(self):
            return StringReducer(self._cols, self._op, self._out or "reduced", self._delim)

    def bind(self, s):
        self._idx = [s.getIndexOfColumn(c) for c in self.inputColumns]
        return super().bind(s)

    def outputSchema(self, s):
        return Schema(list(s.columns) + [ColumnMetaData(self.outputColumnName, ColumnType.String)])

    def map(self, r):
        vals = [r[i].toString() for i in self._idx]
        if self.op == StringReduceOp.APPEND:
            out = "".join(vals)
        elif self.op == StringReduceOp.PREPEND:
            out = "".join(reversed(vals))
        elif self.op == StringReduceOp.MERGE:
            out = self.delimiter.join(vals)
        elif self.op == StringReduceOp.REPLACE:
            out = vals[-1] if vals else ""
        else:
            raise ValueError("unknown StringReduceOp %r" % self.op)
        return list(r) + [Text(out)]


# ----------------------------------------------------------------------------- builder methods
def _install():
    B = TransformProcess.Builder
    B.firstDigitTransform = lambda self, inCol, outCol, mode=FirstDigitTransform.IncludeOtherCategory: self._add(FirstDigitTransform(inCol, outCol, mode))
    B.pivot = lambda self, key, value, default=None: self._add(PivotTransform(key, value, default))
    B.replaceEmptyIntegerWithValue = lambda self, col, v: self._add(ReplaceEmptyIntegerWithValue(col, v))
    B.parseDouble = lambda self: self._add(ParseDoubleTransform())
    B.log2Normalizer = lambda self, col, mean, mn, scale: self._add(Log2Normalizer(col, mean, mn, scale))
    B.mapAllStringsExceptList = lambda self, col, newValue, exceptions: self._add(MapAllStringsExceptList(col, newValue, exceptions))
    B.stringListToCategoricalSet = lambda self, col, names, tokens, delim=",": self._add(StringListToCategoricalSet(col, names, tokens, delim))
    B.stringListToCountsNDArray = lambda self, col, vocab, delim=",", binary=False, ignoreUnknown=False: self._add(
        StringListToCountsNDArray(col, vocab, delim, None, binary, ignoreUnknown))
    B.stringListToIndicesNDArray = lambda self, col, vocab, delim=",", ignoreUnknown=False: self._add(
        StringListToIndicesNDArray(col, vocab, delim, None, False, ignoreUnknown))
    B.textToCharacterIndex = lambda self, col, new, charMap, exceptionOnUnknown=True: self._add(TextToCharacterIndexTransform(col, new, charMap, exceptionOnUnknown))
    B.textToTermIndexSequence = lambda self, col, new, wordMap, delim=" ", exceptionOnUnknown=True: self._add(
        TextToTermIndexSequenceTransform(col, new, wordMap, delim, exceptionOnUnknown))
    B.ndArrayScalarOpTransform = lambda self, col, op, v: self._add(NDArrayScalarOpTransform(col, op, v))
    B.ndArrayMathFunctionTransform = lambda self, col, fn: self._add(NDArrayMathFunctionTransform(col, fn))
    B.ndArrayColumnsMathOpTransform = lambda self, new, op, *cols: self._add(NDArrayColumnsMathOpTransform(new, op, *cols))
    B.ndArrayDistanceTransform = lambda self, new, dist, a, b: self._add(NDArrayDistanceTransform(new, dist, a, b))
    B.sequenceDifference = lambda self, col, new=None, lookback=1, mode=SequenceDifferenceTransform.Default, value=None: self._add(
        SequenceDifferenceTransform(col, new, lookback, mode, value))
    B.trimSequenceToLength = lambda self, maxLength, mode=SequenceTrimToLength.Trim, pad=None: self._add(SequenceTrimToLength(maxLength, mode, pad))
    B.trimOrPadSequenceToLength = lambda self, maxLength, pad: self._add(SequenceTrimToLength(maxLength, SequenceTrimToLength.TrimOrPad, pad))
    B.reduceSequence = lambda self, reducer: self._add(ReduceSequenceTransform(reducer))
    B.reduceSequenceByWindow = lambda self, reducer, window: self._add(ReduceSequenceByWindowTransform(reducer, window))
    B.splitSequenceByTimeSeparation = lambda self, col, quantity, unit="MILLISECONDS": self._add(SequenceSplitTimeSeparation(col, quantity, unit))
    B.stringReduce = lambda self, reducer: self._add(reducer)
    B.filterInvalidNumColumns = lambda self: self._add(InvalidNumColumns())


_install()
